# round 4, call k: the default bench run (as the driver runs it) + the HIP API trace of one `dipper` command
O=gpurun_out/r4/k; mkdir -p $O
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; cat $O/bench_time.txt | tail -3
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup", "data")})
print("roofline", d["roofline"])
print("hot_path phase_ms", d["phase_ms"], "bench_wall_s", d.get("bench_wall_s"))
print("cpu_baseline", d.get("cpu_baseline"))
print("nj_iteration_scaling", d.get("nj_iteration_scaling"))
oc = d.get("other_configs", {})
for k, v in oc.items():
    print(k, {kk: v.get(kk) for kk in ("seconds", "tips_per_s", "queries_per_s", "nj_ms", "nrf_vs_generating_tree", "phases_ms", "skipped", "error", "leg_wall_s")})
PY
tools/bin/gen_synth --tips 30000 --sites 10000 --seed 1 --indel-gaps --fasta /dev/shm/r4k.fa
( cd /tmp && export TMPDIR=/tmp DPR_CLI_NORMAL_EXIT=1 && rocprofv3 --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/hiptrace -o t -- $GRAFT_REPO_ROOT/dipper_amd/bin/dipper -i m -I /dev/shm/r4k.fa -O /dev/shm/r4k.nwk -m 2 -d 2 > $GRAFT_REPO_ROOT/$O/hiptrace.out 2> $GRAFT_REPO_ROOT/$O/hiptrace.err )
S=$(find $O/hiptrace -name "*hip_api_stats.csv" | head -1); [ -n "$S" ] && head -16 $S | tee $O/cli_hip_api_stats.csv
find $O/hiptrace -name "*_trace.csv" -size +5M -delete
rm -f /dev/shm/r4k.fa /dev/shm/r4k.nwk
