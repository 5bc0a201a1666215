#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
timeout -k 10 300 python -m pytest tests/test_gpu_exact.py tests/test_gpu_cli.py -m gpu -x -q > $OUT/tests_exact2.out 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests_exact2.out
timeout -k 10 100 python profiles/exact_bench.py 10000 2000
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exact30k -o v -- python3 $REPO/profiles/exact_bench.py 30000 2000 > $OUT/exact30k_prof.out 2> $OUT/exact30k_prof.err
cat $OUT/exact30k_prof.out
find $OUT/exact30k -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/exact30k/**/v_kernel_stats.csv", recursive=True) + glob.glob("$OUT/exact30k/v_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:9]:
        print(r['Name'][:64].ljust(64), r['Calls'].rjust(7), "%10.1f ms" % (float(r['TotalDurationNs'])/1e6), "%9.2f us avg" % (float(r['AverageNs'])/1e3), r['Percentage'])
    break
PY
cd $REPO && timeout -k 10 200 python profiles/exact_bench.py 100000 1000
