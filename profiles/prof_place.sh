#!/bin/bash
# usage (GPU box): bash profiles/prof_place.sh <tag> <tips> <sites> <kind>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 $GRAFT_REPO_ROOT/profiles/place_bench.py "$@" > $OUT/out.txt 2>&1
rm -f $OUT/*kernel_trace.csv
python3 - <<PY
import csv
for r in csv.DictReader(open('$OUT/b_kernel_stats.csv')):
    print(r['Name'][:56].ljust(56), r['Calls'].rjust(7), "%10.1f ms"%(float(r['TotalDurationNs'])/1e6), "%9.2f us avg"%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
tail -1 $OUT/out.txt
