#!/bin/bash
# usage (GPU box): bash profiles/pmc_kernel.sh <tag> "<counters>" <kernel substring> -- <python script args...>
# one rocprofv3 --pmc pass (with --kernel-trace only); prints the per-dispatch mean of every counter for the kernel
TAG=$1; CNT=$2; KSUB=$3; shift 3; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -o p -- python3 "$@" > $OUT/out.txt 2>&1
python3 - <<PY
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open('$OUT/p_counter_collection.csv')):
    if '$KSUB' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print('$TAG', k, len(v), 'mean %.4e'%(sum(v)/len(v)))
PY
rm -f $OUT/p_kernel_trace.csv
