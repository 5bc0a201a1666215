#!/bin/bash
# usage (GPU box): bash profiles/prof_trace.sh <tag> [bench args]  -> keeps the gzipped kernel trace
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
gzip -9 -f $OUT/b_kernel_trace.csv
ls -la $OUT
