"""Whole-command wall time of `dipper` (30 000 x 10 000 FASTA -> Newick, back to back) under HIP / ROCr environment settings that change
how many queues the runtime creates and tears down: python profiles/cli_env_sweep.py [tips] [sites] [runs]"""
import json, os, statistics, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tmp = tempfile.mkdtemp(prefix="cliev_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fa = os.path.join(tmp, "a.fa")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1", "--fasta", fa], check=True)
exe = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
D = {"DPR_CLI_RUNTIME_DEFAULTS": "1"}       # (main.cpp sets HSA_ENABLE_SDMA=0 and GPU_MAX_HW_QUEUES=2 for itself unless this is set)
variants = [("runtime defaults", D), ("HSA_ENABLE_SDMA=0 GPU_MAX_HW_QUEUES=2 (what the command sets for itself)", {}),
            ("HSA_ENABLE_SDMA=0 only", dict(D, HSA_ENABLE_SDMA="0")), ("GPU_MAX_HW_QUEUES=2 only", dict(D, GPU_MAX_HW_QUEUES="2"))] * 2
if os.environ.get("SWEEP_EXIT"):
    variants = [("_exit with everything live (DPR_CLI_FAST_EXIT=1)", {"DPR_CLI_FAST_EXIT": "1"}), ("dpr_destroy, then _exit (default)", {})] * 3
for name, env in variants:
    walls, mains, inputs = [], [], []
    for r in range(runs + 1):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-i", "m", "-I", fa, "-O", os.path.join(tmp, "o.nwk"), "-m", "2", "-d", "2"], capture_output=True, text=True,
                           env=dict(os.environ, DPR_CLI_TIMING="1", **env))
        wall = (time.perf_counter() - t0) * 1e3
        if p.returncode != 0:
            print(name, "FAILED", p.stderr[-300:]); break
        if r == 0:
            continue            # first run of a variant: the previous variant's teardown
        m = [float(l.split(":")[1].split()[0]) for l in p.stderr.splitlines() if l.startswith("Main in:")]
        i = [float(l.split(":")[1].split()[0]) for l in p.stderr.splitlines() if l.startswith("Input in:")]
        walls.append(wall); mains.append(m[0] if m else -1); inputs.append(i[0] if i else -1)
    if walls:
        print(json.dumps({"variant": name, "wall_ms_median": round(statistics.median(walls), 1), "wall_ms": [round(w) for w in walls],
                          "main_ms_median": statistics.median(mains), "input_ms_median": statistics.median(inputs)}), flush=True)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
