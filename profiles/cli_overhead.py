"""Where the `dipper` command's wall time goes outside its own clock: python profiles/cli_overhead.py [tips] [sites] [runs]
wall (parent's clock around the child) vs "Main in" (the child's clock from main() to the exit call): the difference is process start
(exec, dynamic linking, static initialisers) + process end (the kernel driver tearing down the GPU context and its allocations)."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
tmp = tempfile.mkdtemp(prefix="cliov_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fa = os.path.join(tmp, "a.fa")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1", "--fasta", fa], check=True)
exe = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
out = []
for mode in ("plain", "help", "true"):
    for r in range(runs):
        cmd = [exe, "-i", "m", "-I", fa, "-O", os.path.join(tmp, "o.nwk"), "-m", "2", "-d", "2"] if mode == "plain" else [exe, "--help"] if mode == "help" else ["/bin/true"]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, DPR_LOG="cli"))
        wall = (time.perf_counter() - t0) * 1e3
        main_ms = None
        for line in p.stderr.splitlines():
            if line.startswith("Main in:"):
                main_ms = float(line.split(":")[1].split()[0])
        out.append({"mode": mode, "wall_ms": round(wall, 1), "main_ms": main_ms, "outside_ms": None if main_ms is None else round(wall - main_ms, 1)})
for o in out:
    print(json.dumps(o))
import shutil
shutil.rmtree(tmp, ignore_errors=True)
