#!/bin/bash
# Run ON THE GPU BOX through gpurun: collects the rocprofv3 summaries committed under profiles/.
#   $1 = tag (e.g. r1)
set -u
TAG=${1:-r1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1) kernel trace + stats of the default bench command (1 timed step to bound the trace size)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# 2) PMC passes on the probe only (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o probe -- python3 $REPO/bench.py --probe-only --no-cpu-baseline > $OUT/probe_fetch.json 2> $OUT/probe_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o probe -- python3 $REPO/bench.py --probe-only --no-cpu-baseline > $OUT/probe_write.json 2> $OUT/probe_write.err
find $OUT -name "*.csv" | head -20
# keep only what fits the 64 MiB merge budget: stats + compact PMC rows of the scan kernel
for f in $(find $OUT/stats -name "*kernel_trace.csv"); do gzip -9 $f; done
ls -la $OUT/*/* | head -30
