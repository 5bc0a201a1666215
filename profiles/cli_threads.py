"""CLI input phase against the host threads a rank gets (8 ranks on a 16-core grant: 2 each): python profiles/cli_threads.py"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
tmp = tempfile.mkdtemp(prefix="clit_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fa = os.path.join(tmp, "in.fa")
subprocess.run([bench.GEN, "--tips", "30000", "--sites", "10000", "--seed", "1", "--fasta", fa], check=True)
for th in (16, 4, 2, 1):
    rows = []
    for rep in range(5):
        dt, ph = bench.cli_step(fa, os.path.join(tmp, "out.nwk"), 0, th)
        rows.append((dt * 1e3, ph.get("input"), ph.get("device_ready"), ph.get("tree")))
        time.sleep(0.3)
    rows = rows[1:]
    print(json.dumps({"host_threads": th, "wall_ms": [round(r[0]) for r in rows], "input_ms": [r[1] for r in rows],
                      "device_ready_ms": [r[2] for r in rows], "tree_ms": [r[3] for r in rows]}), flush=True)
import shutil; shutil.rmtree(tmp, ignore_errors=True)
