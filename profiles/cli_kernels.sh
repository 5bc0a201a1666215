#!/bin/bash
# usage (GPU box): bash profiles/cli_kernels.sh  -> per-kernel rocprofv3 stats of ONE CLI run at 30 000 x 10 000 (are the NJ kernels of a fresh process slower, or are there gaps?)
python3 - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from tests import _util
import bench
seqs = bench.make_input(30000, 10000, 1)
_util.write_fasta("/tmp/in.fa", ["T%d" % (i + 1) for i in range(30000)], seqs, width=0)
PY
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cli
mkdir -p $OUT
EXE=$GRAFT_REPO_ROOT/dipper_amd/bin/dipper
cd /tmp && export TMPDIR=/tmp
DPR_CLI_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c -- $EXE -i m -I /tmp/in.fa -O /tmp/o.nwk -m 2 -d 2 > $OUT/out.txt 2> $OUT/err.txt
grep -E "device:|Main in" $OUT/err.txt $OUT/out.txt | head -3
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$OUT/c_kernel_stats.csv')))[:6]:
    print(r['Name'][:50].ljust(50), r['Calls'].rjust(7), "%9.1f ms"%(float(r['TotalDurationNs'])/1e6), "%8.2f us avg"%(float(r['AverageNs'])/1e3))
import collections
rows=list(csv.DictReader(open('$OUT/c_kernel_trace.csv')))
nj=[r for r in rows if 'njp_' in r['Kernel_Name']]
t0=min(int(r['Start_Timestamp']) for r in nj); t1=max(int(r['End_Timestamp']) for r in nj)
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in nj)
print("njp kernels: span %.1f ms, busy %.1f ms, %d launches"%((t1-t0)/1e6, busy/1e6, len(nj)))
PY
rm -f $OUT/c_kernel_trace.csv
