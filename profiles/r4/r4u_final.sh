# round 4, call u: the driver's two commands at HEAD -- the whole -m gpu suite, then bench.py --gpus 1 --steps 20 --warmup 5
O=gpurun_out/r4/u; mkdir -p $O
( time python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log; tail -3 $O/pytest_time.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; tail -3 $O/bench_time.txt
python3 - <<PY
import json
d = json.load(open("$O/bench_steps20_warmup5.json"))
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "steps", "warmup", "dtype", "scaling", "vs_baseline")})
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "frac", "traffic")})
print("cpu_baseline", d.get("cpu_baseline"))
print("cpu_baseline_rapidnj", d.get("cpu_baseline_rapidnj"))
print("phase_ms", d["phase_ms"], "bench_wall_s", d.get("bench_wall_s"), "step_ms", d["step_ms"])
for k, v in d.get("other_configs", {}).items():
    print(k, {kk: v.get(kk) for kk in ("seconds", "addquery_s", "tips_per_s", "queries_per_s", "nj_ms", "skipped", "error")})
PY
