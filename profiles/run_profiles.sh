#!/bin/bash
# Run ON THE GPU BOX through gpurun: collects the rocprofv3 summaries that are committed under profiles/<tag>/.
#   bash profiles/run_profiles.sh r2
# 1) kernel trace + stats of the hot path of the default bench workload (one in-process step: the CLI steps, the parity
#    legs and the CPU baselines are left out so that the summary holds the kernels of the timed path only);
# 2) PMC passes on the Q-argmin probe only (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together);
# 3) kernel stats of NJ at 100 000 x 10 000.
set -u
TAG=${1:-r2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cli --no-parity --no-cpu-baseline --no-other-configs --no-stream-leg > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o probe -- python3 $REPO/bench.py --probe-only --no-cli --no-parity --no-cpu-baseline --no-other-configs --no-stream-leg > $OUT/probe_fetch.json 2> $OUT/probe_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o probe -- python3 $REPO/bench.py --probe-only --no-cli --no-parity --no-cpu-baseline --no-other-configs --no-stream-leg > $OUT/probe_write.json 2> $OUT/probe_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats100k -o bench -- python3 $REPO/bench.py --tips 100000 --sites 10000 --steps 1 --warmup 0 --no-cli --no-parity --no-cpu-baseline --no-other-configs --no-stream-leg > $OUT/bench100k_under_rocprof.json 2> $OUT/bench100k_under_rocprof.err
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
for tag, pat in (("30k", "$OUT/stats/*kernel_stats.csv"), ("100k", "$OUT/stats100k/*kernel_stats.csv")):
    for f in glob.glob(pat):
        print("==", tag)
        for r in list(csv.DictReader(open(f)))[:8]:
            print(r['Name'][:52].ljust(52), r['Calls'].rjust(7), "%10.1f ms"%(float(r['TotalDurationNs'])/1e6), "%9.2f us avg"%(float(r['AverageNs'])/1e3), r['Percentage'])
for kind in ("fetch", "write"):
    for f in glob.glob("$OUT/pmc_%s/*counter_collection.csv" % kind):
        v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'nj_scan_kernel' in r['Kernel_Name']]
        if v: print(kind, "nj_scan_kernel probe:", len(v), "dispatches, mean counter", sum(v)/len(v))
PY
