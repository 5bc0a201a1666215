#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exact30k -o v -- python3 $REPO/profiles/exact_bench.py 30000 2000 > $OUT/exact30k_prof.out 2> $OUT/exact30k_prof.err
cat $OUT/exact30k_prof.out
find $OUT/exact30k -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/exact30k/**/v_kernel_stats.csv", recursive=True) + glob.glob("$OUT/exact30k/v_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r['Name'][:64].ljust(64), r['Calls'].rjust(7), "%10.1f ms" % (float(r['TotalDurationNs'])/1e6), "%9.2f us avg" % (float(r['AverageNs'])/1e3), r['Percentage'])
    break
PY
