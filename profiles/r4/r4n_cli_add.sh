# round 4, call n: --add through the packed reader -- CLI parity tests, then the default bench run (add legs)
O=gpurun_out/r4/n; mkdir -p $O
python -m pytest tests/test_gpu_cli.py -x -q -m gpu > $O/test_gpu_cli.log 2>&1; rc=$?; echo "cli tests rc=$rc"; tail -3 $O/test_gpu_cli.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/test_gpu_cli.log | head -20; exit 1; fi
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["phase_ms"], d["roofline"]["frac"], d.get("bench_wall_s"))
print(d["e2e_cli"]["wall_ms"], d["e2e_cli"]["hip_startup_ms"])
print("rapidnj", d.get("cpu_baseline_rapidnj"))
for k, v in d.get("other_configs", {}).items():
    print(k, {kk: v.get(kk) for kk in ("seconds", "addquery_s", "tips_per_s", "queries_per_s", "nj_ms", "phases_ms", "skipped", "error", "leg_wall_s")})
PY
