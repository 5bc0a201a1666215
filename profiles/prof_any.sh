#!/bin/bash
# usage (GPU box): bash profiles/prof_any.sh <tag> <script.py> [args...]  -> per-kernel stats of any helper script
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 $SCRIPT "$@" > $OUT/out.txt 2> $OUT/err.txt
rm -f $OUT/*kernel_trace.csv
python3 - <<PY
import csv
for r in csv.DictReader(open('$OUT/b_kernel_stats.csv')):
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(8), "%10.1f ms"%(float(r['TotalDurationNs'])/1e6), "%9.2f us avg"%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
tail -2 $OUT/out.txt
