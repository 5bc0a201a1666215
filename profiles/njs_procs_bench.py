"""Per-iteration cost of the one-exchange row-sharded NJ loop with the ranks as PROCESSES on ONE GPU (mailbox plan: IPC
windows, no RCCL -- it refuses two ranks on one device).  The ranks share the GPU's HBM, so the scan part of an iteration
cannot get faster than the single-process streaming loop; what the figure shows is the OVERHEAD the exchange adds to it:
    overhead = us/iteration (W processes) - us/iteration (1 process, nj.hip streaming loop), same input, same iterations.
  python profiles/njs_procs_bench.py [tips] [sites] [iters] [worlds, e.g. 2,4]
Worker mode (internal): ... --worker rank world p4 tips sites iters out.json"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def worker(rank, world, p4, n, L, iters, out):
    import dipper_amd
    from dipper_amd import capi
    packed = np.memmap(p4, dtype=np.uint64, mode="r", shape=(n, (L + 15) // 16))
    d = dipper_amd.Dipper(0)
    d.set_nj_mode(0)
    if world > 1:
        d.comm_init_local(rank, world)
        blob = d.peer_export(n)
        sys.stdout.write(blob.hex() + "\n"); sys.stdout.flush()
        blobs = [bytes.fromhex(sys.stdin.readline().strip()) for _ in range(world)]
        d.peer_attach(blobs)
    d.set_msa(packed, L)
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    d.nj_run(max_iters=8)
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    sys.stdout.write("ready\n"); sys.stdout.flush()
    sys.stdin.readline()                      # go
    t0 = time.perf_counter()
    res = d.nj_run(max_iters=iters)
    wall = time.perf_counter() - t0
    _, nj_ms = d.timing()
    info = d.nj_exchange_info() if world > 1 else {"plan": "single process (nj.hip)", "launches": 2 * iters, "collectives": 0, "note": ""}
    import hashlib
    h = hashlib.sha256()
    for k in ("merge_x", "merge_y", "bl_x", "bl_y"):
        h.update(np.ascontiguousarray(res[k][:res["iters"]]).tobytes())
    json.dump({"rank": rank, "world": world, "iters": int(res["iters"]), "wall_s": wall, "loop_ms_events": nj_ms,
               "us_per_iteration": nj_ms * 1e3 / max(res["iters"], 1), "digest": h.hexdigest()[:16], **info}, open(out, "w"))
    d.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    worlds = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1,2,4").split(",")]
    tmp = tempfile.mkdtemp(prefix="njsb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        p4 = os.path.join(tmp, "a.p4")
        subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1", "--packed4", p4], check=True)
        base = None
        for W in worlds:
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(W), p4, str(n), str(L), str(iters),
                                       os.path.join(tmp, "o%d_%d.json" % (W, r))], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
                     for r in range(W)]
            try:
                if W > 1:
                    blobs = [p.stdout.readline().strip() for p in procs]
                    for p in procs:
                        p.stdin.write("\n".join(blobs) + "\n"); p.stdin.flush()
                for p in procs:
                    assert p.stdout.readline().strip() == "ready"
                for p in procs:
                    p.stdin.write("go\n"); p.stdin.flush()
                for p in procs:
                    p.wait(timeout=600)
            finally:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            recs = [json.load(open(os.path.join(tmp, "o%d_%d.json" % (W, r)))) for r in range(W)]
            us = max(r["us_per_iteration"] for r in recs)
            if W == 1:
                base = us
            print(json.dumps({"tips": n, "sites": L, "iterations": iters, "processes_on_one_gpu": W, "plan": recs[0]["plan"],
                              "us_per_iteration": us, "overhead_us_vs_single_process": None if base is None or W == 1 else us - base,
                              "launches_per_iteration": recs[0]["launches"] / max(recs[0]["iters"], 1),
                              "collectives_per_iteration": recs[0]["collectives"] / max(recs[0]["iters"], 1),
                              "digests_equal": len({r["digest"] for r in recs}) == 1, "digest": recs[0]["digest"]}), flush=True)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        a = sys.argv[2:]
        worker(int(a[0]), int(a[1]), a[2], int(a[3]), int(a[4]), int(a[5]), a[6])
    else:
        main()
