#!/bin/bash
# round 3: default bench run on the final tree
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
t0=$(date +%s)
timeout -k 10 600 python bench.py > $OUT/bench_default_final.json 2> $OUT/bench_default_final.err
echo "rc=$? wall=$(( $(date +%s) - t0 ))s"
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_default_final.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "wall", d.get("bench_wall_s"))
print("hot", json.dumps(d.get("hot_path", {}).get("phases_ms"))[:300], d.get("hot_path", {}).get("ms_per_step"))
print("roofline", d.get("roofline"))
print("cpu", d.get("cpu_baseline"))
oc = d.get("other_configs", {})
for k, v in oc.items():
    print(k, json.dumps(v)[:400])
PY
