#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
for v in 0 1; do
  if [ $v = 1 ]; then export DPR_NJP_PERMUTE_2D=1; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/perm$v -o b -- python3 $REPO/bench.py --tips 100000 --sites 10000 --steps 1 --warmup 0 --no-cli --no-parity --no-cpu-baseline --no-other-configs --no-stream-leg > $OUT/perm$v.json 2> $OUT/perm$v.err
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/perm$v/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "permute" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("variant $v:", rows[0]["Kernel_Name"][:40], len(d), "calls; ms:", [round(x, 1) for x in d[:14]])
PY
  rm -rf $OUT/perm$v
done
