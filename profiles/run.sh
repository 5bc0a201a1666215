#!/bin/bash
# ONE evidence driver for the GPU box (with profiles/prof.sh for the rocprofv3 passes): every record under profiles/r<round>/ was
# produced by one of these steps (profiles/r<round>/README.md maps record -> step).  Steps are joined with && inside one gpurun call:
#   bash profiles/run.sh suite                     whole `-m gpu` suite as the driver runs it, then smoke()
#   bash profiles/run.sh tests <tag> <pytest args> a subset, first failure lines printed, non-zero exit on failure
#   bash profiles/run.sh bench <tag> [bench args]  bench.py, its JSON line kept, headline fields printed
#   bash profiles/run.sh variants <tag> "<nj_target.py args>" [VAR=VALUE|-]...   NJ timing per env variant ("-" = none)
#   bash profiles/run.sh rehearse                  bench.py's multi-rank control flow on ONE GPU (1 rank RCCL; 2 process ranks)
#   bash profiles/run.sh hiptrace <tag>            HIP API trace of one `dipper` command (30 000 x 10 000)
# Output under gpurun_out/${DPR_ROUND:-r5}/<tag>/.
set -u
STEP=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
case $STEP in
suite)
    O=gpurun_out/${DPR_ROUND:-r5}/suite; mkdir -p $O
    ( time python -m pytest tests/ -x -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_time.txt; rc=$?
    echo "pytest rc=$rc"; tail -22 $O/pytest_gpu.log; tail -3 $O/pytest_time.txt
    [ $rc -eq 0 ] || exit 1
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
    ;;
tests)
    TAG=$1; shift; O=gpurun_out/${DPR_ROUND:-r5}/$TAG; mkdir -p $O
    python -m pytest "$@" -x -q -m gpu > $O/tests.log 2>&1; rc=$?
    echo "tests [$*] rc=$rc"; tail -3 $O/tests.log
    if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head -20; exit 1; fi
    ;;
bench)
    TAG=$1; shift; O=gpurun_out/${DPR_ROUND:-r5}/$TAG; mkdir -p $O
    ( time python bench.py "$@" > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; rc=$?
    echo "bench rc=$rc"; tail -3 $O/bench_time.txt
    [ $rc -eq 0 ] || { tail -5 $O/bench.err; exit 1; }
    python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "steps", "warmup", "dtype", "scaling", "vs_baseline")})
print("roofline", d["roofline"])
print("cpu_baseline", d.get("cpu_baseline")); print("cpu_baseline_rapidnj", d.get("cpu_baseline_rapidnj"))
print("phase_ms", d.get("phase_ms"), "bench_wall_s", d.get("bench_wall_s"))
print("nj_iteration_scaling", json.dumps(d.get("nj_iteration_scaling"))[:1500])
for k, v in d.get("other_configs", {}).items():
    print(k, {kk: v.get(kk) for kk in ("seconds", "addquery_s", "tips_per_s", "queries_per_s", "nj_ms", "phases_ms", "skipped", "error", "leg_wall_s")})
PY
    ;;
variants)
    TAG=$1; ARGS=$2; shift 2; O=gpurun_out/${DPR_ROUND:-r5}/$TAG; mkdir -p $O
    [ $# -gt 0 ] || set -- -
    for v in "$@"; do
        [ "$v" = "-" ] && v=""
        echo "== nj_target.py $ARGS [$v]" | tee -a $O/variants.txt
        env $v python3 profiles/nj_target.py --no-torch $ARGS 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants.txt
    done
    ;;
rehearse)
    O=gpurun_out/${DPR_ROUND:-r5}/rehearse; mkdir -p $O
    export HSA_ENABLE_IPC_MODE_LEGACY=0
    DPR_BENCH_CHECK=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 > $O/bench_check_one_rank_rccl.json 2> $O/a.err; rc=$?
    echo "one rank, RCCL, cross-checks on: rc=$rc"; [ $rc -eq 0 ] || { tail -5 $O/a.err; exit 1; }
    DPR_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_2proc_one_gpu.json 2> $O/b.err; rc=$?
    echo "two process ranks on one GPU: rc=$rc"; [ $rc -eq 0 ] || { tail -5 $O/b.err; exit 1; }
    python3 - <<PY
import json
for f in ("bench_check_one_rank_rccl", "bench_2proc_one_gpu"):
    d = json.load(open("$O/%s.json" % f))
    print(f, d["value"], d["n_gpus"], d.get("bench_wall_s"), json.dumps(d.get("nj_iteration_scaling"))[:1200])
PY
    ;;
hiptrace)
    TAG=$1; O=gpurun_out/${DPR_ROUND:-r5}/$TAG; mkdir -p $O
    tools/bin/gen_synth --tips 30000 --sites 10000 --seed 1 --indel-gaps --fasta /dev/shm/ht.fa
    for k in 1 2 3; do dipper_amd/bin/dipper -i m -I /dev/shm/ht.fa -O /dev/shm/ht.nwk -m 2 -d 2 2>&1 | grep -E "Device ready|Input in|Tree Created" | tr '\n' ' '; echo; done | tee $O/cli_startup_plain.txt
    ( cd /tmp && export TMPDIR=/tmp DPR_CLI_NORMAL_EXIT=1 && rocprofv3 --hip-trace --stats --output-format csv -d $R/$O/hiptrace -o t -- $R/dipper_amd/bin/dipper -i m -I /dev/shm/ht.fa -O /dev/shm/ht.nwk -m 2 -d 2 > $R/$O/hiptrace.out 2> $R/$O/hiptrace.err )
    S=$(find $O/hiptrace -name "*hip_api_stats.csv" | head -1); [ -n "$S" ] && head -16 $S | tee $O/cli_hip_api_stats.csv
    find $O/hiptrace -name "*_trace.csv" -size +5M -delete
    rm -f /dev/shm/ht.fa /dev/shm/ht.nwk
    ;;
*) echo "unknown step $STEP"; exit 2;;
esac
