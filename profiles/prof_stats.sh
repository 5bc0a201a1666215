#!/bin/bash
# usage (on the GPU box): bash profiles/prof_stats.sh <tag> [bench args...]   -> prints per-kernel stats of the hot path
# (in-process steps only: no CLI steps, no parity legs, no CPU baselines); summary kept as gpurun_out/prof_<tag>/b_kernel_stats.csv
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cli --no-parity --no-cpu-baseline --no-other-configs --no-stream-leg "$@" > $OUT/bench.json 2> $OUT/bench.err
rm -f $OUT/*kernel_trace.csv
python3 - <<PY
import csv
for r in csv.DictReader(open('$OUT/b_kernel_stats.csv')):
    print(r['Name'][:48].ljust(48), r['Calls'].rjust(7), "%10.1f ms"%(float(r['TotalDurationNs'])/1e6), "%9.2f us avg"%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['phase_ms'], d['hot_path'].get('prune'))"
