#!/bin/bash
# usage (GPU box): bash profiles/pmc_njp.sh <tag> "<counters>" [iterations 1500] [tips 30000]   -> one rocprofv3 --pmc pass (kernel trace only) over one
# 30 000 x 10 000 NJ run, first 1 500 iterations (eager launches: DPR_NJ_NOGRAPH=1, so every dispatch is its own record); prints the per-dispatch
# mean of every counter for the two kernels of the pruned loop as CSV lines "tag,kernel,counter,dispatches,mean"
TAG=$1; CNT=$2; ITERS=${3:-1500}; TIPS=${4:-30000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DPR_NJ_NOGRAPH=1
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/profiles/njp_pmc_target.py $ITERS $TIPS > $OUT/out.txt 2>&1
python3 - <<PY
import csv, collections, glob
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/**/p_counter_collection.csv', recursive=True) + glob.glob('$OUT/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for k in ('njp_scan_kernel', 'njp_post_kernel', 'njp_post2_kernel'):
            if k in r['Kernel_Name']:
                acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    break
for (k, c), v in sorted(acc.items()):
    print('$TAG,%s,%s,%d,%.6e' % (k, c, len(v), sum(v) / len(v)))
PY
tail -2 $OUT/out.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
