"""ONE parameterised target for everything that profiles the NJ loop (replaces the one-off scripts of earlier rounds):

  python3 profiles/nj_target.py [--tips 30000] [--sites 10000] [--seed 1] [--iters -1] [--reps 1] [--mode pruned|stream]
                                [--phases ITERATION] [--json out.json]

Generates the alignment with tools/bin/gen_synth (the bench's input for seed 1), builds the JC69 matrix and runs the NJ loop
`reps` times; prints one JSON line per repetition: nj_ms, microseconds per iteration, units listed, digest of the merge log.
--phases: DPR_NJ_PHASES stamps of that iteration (eager launches of it), summarised per kernel and role.
Run it as `rocprofv3 ... -- python3 /abs/path/profiles/nj_target.py ...` -- the interpreter itself behind `--` (no shell,
no `env`, no shebang hop: the profiler's preload has initialised the GPU by then, and an exec from such a process is refused on
the GPU boxes); profiles/prof.sh does exactly that."""
import argparse, hashlib, json, os, subprocess, sys, tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--tips", type=int, default=30000)
ap.add_argument("--sites", type=int, default=10000)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--gap-frac", type=float, default=0.0)
ap.add_argument("--indel-gaps", action="store_true")
ap.add_argument("--model", default="jc69", help="substitution model of the generated alignment: jc69 | gtr+g+i (gen_synth --model)")
ap.add_argument("--bl-scale", type=float, default=1.0, help="branch lengths of the generating tree times this (1 = the authors' -rlen; 20: hardly any identical tips, so hardly any tied q)")
ap.add_argument("--iters", type=int, default=-1)
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--mode", default="pruned")
ap.add_argument("--phases", type=int, default=None)
ap.add_argument("--json", default=None)
ap.add_argument("--clocks", type=float, default=0.0, help="seconds into the NJ loop at which `rocm-smi --showclocks` is sampled (0 = not)")
ap.add_argument("--spin", type=int, default=0, help="background load while the NJ loop runs: this many 64-thread workgroups spinning on FMAs (dpr_spin_start)")
ap.add_argument("--no-torch", action="store_true",
                help="do not import torch first.  With torch imported the process runs on the HIP runtime bundled with PyTorch (7.0), as bench.py "
                     "does; without it on /opt/rocm's 7.2 -- where rocprofv3 --kernel-trace segfaults inside the first hipGraphLaunch of the "
                     "pruned loop (round 4, gpurun_out/r4/base30k/err.txt); eager launches (DPR_NJ_NOGRAPH=1) and unprofiled runs are fine there")
args = ap.parse_args()
if not args.no_torch:
    import torch  # noqa: F401
if args.phases is not None:
    os.environ["DPR_NJ_PHASES"] = str(args.phases)

import numpy as np  # noqa: E402
import dipper_amd  # noqa: E402
from dipper_amd import capi  # noqa: E402


def alignment(n, L, seed, gap=0.0, indel_gaps=False, bl_scale=1.0, model="jc69"):
    k = bl_scale * 10000.0 / L            # same expected number of substitutions per branch as the bench's 10 000 sites
    tmp = tempfile.mkdtemp(prefix="njt_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    p4 = os.path.join(tmp, "a.p4")
    subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", str(seed),
                    "--mean-bl", repr(2e-5 * k), "--lo", repr(2e-6 * k), "--hi", repr(2e-4 * k), "--packed4", p4]
                   + (["--gap-frac", repr(gap)] if gap > 0 else []) + (["--indel-gaps"] if indel_gaps else []) + (["--model", model] if model != "jc69" else []), check=True)
    packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
    os.unlink(p4); os.rmdir(tmp)
    return packed


def digest(res):
    h = hashlib.sha256()
    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
        h.update(np.ascontiguousarray(res[key]).tobytes())
    return h.hexdigest()[:16]


packed = alignment(args.tips, args.sites, args.seed, args.gap_frac, args.indel_gaps, args.bl_scale, args.model)
capi.set_nj_mode(0 if args.mode == "stream" else 1)
d = dipper_amd.Dipper(0)
d.set_msa(packed, args.sites)
out = []
for rep in range(args.reps):
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    if args.spin > 0:
        import ctypes as C
        L_ = capi.load_library()
        L_.dpr_spin_start.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L_.dpr_spin_stop.argtypes = [C.c_void_p]
        assert L_.dpr_spin_start(d.h, args.spin, 20000) == 0
    clocks = {}
    if args.clocks:
        import threading, time as _t
        def _peek():
            _t.sleep(args.clocks)
            r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True)
            clocks["during_nj"] = [l.strip() for l in r.stdout.splitlines() if "clk" in l.lower() and "GPU[0]" in l]
        th = threading.Thread(target=_peek); th.start()
    res = d.nj_run(max_iters=args.iters)
    if args.clocks:
        th.join()
    if args.spin > 0:
        assert L_.dpr_spin_stop(d.h) == 0
    dist_ms, nj_ms = d.timing()
    rec = {"tips": args.tips, "sites": args.sites, "gap_frac": args.gap_frac, "indel_gaps": args.indel_gaps, "bl_scale": args.bl_scale, "model": args.model, "spin_blocks": args.spin, "mode": args.mode, "rep": rep, "iters": int(res["iters"]), "dist_ms": dist_ms, "nj_ms": nj_ms,
           "us_per_iteration": nj_ms * 1e3 / max(int(res["iters"]), 1), "units_listed": d.prune_stats()[0] if args.mode != "stream" else None, "digest": digest(res),
           "env": {k: v for k, v in os.environ.items() if k.startswith("DPR_")}}
    if clocks:
        rec["clocks"] = clocks
    out.append(rec)
    print(json.dumps(rec), flush=True)
if args.json:
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)

if args.phases is not None:
    import ctypes as C
    buf = np.zeros(4 * 2048 * 8, np.uint64)
    L_ = capi.load_library()
    L_.dpr_get_nj_phase_stamps.argtypes = [C.c_void_p]
    assert L_.dpr_get_nj_phase_stamps(buf.ctypes.data) == 0
    fine = buf[2 * 2048 * 8:3 * 2048 * 8].reshape(2048, 8).astype(np.int64)      # njp_post2_kernel's finer stamps (group 2)
    buf = buf[:2 * 2048 * 8].reshape(2, 2048, 8).astype(np.int64)
    t_scan0 = buf[0][buf[0] > 0].min()
    for k, name in ((0, "scan"), (1, "post")):
        b = buf[k]
        used = b[:, 0] > 0
        if not used.any():
            print(name, "no stamps")
            continue
        t0 = b[used][:, 0].min()
        print(f"== {name}: {used.sum()} blocks stamped; first stamp {10 * (t0 - t_scan0)} ns after the scan's first; times in ns after this kernel's first stamp")
        roles = [("all", used)]
        if k == 1:
            code = b[:, 7]
            if (code[used] > 0).any() and (code[used] < 16).all():      # njp_post2 / post3: 1 = UM (U) block, 2 = M block, 3 = T block that left after the coarse test, 4 = T block
                roles = [("U / UM", used & (code == 1)), ("M", used & (code == 2)), ("T-coarse-exit", used & (code == 3)), ("T-full", used & (code == 4)), ("other", used & (code == 0))]
            else:                                                        # fused kernel: test blocks stamp slot 5
                roles = [("test", used & (b[:, 5] > 0)), ("update", used & (b[:, 5] == 0))]
        for rname, m in roles:
            if not m.any():
                continue
            print(f"  -- {rname}: {int(m.sum())} blocks")
            for j in range(7 if k == 1 else 8):
                col = b[m][:, j]
                ok = col > 1000000          # (stamps are clock values; some slots of the scan hold small counters)
                if ok.any():
                    v = 10 * (col[ok] - t0)
                    print(f"   stamp {j}: blocks {ok.sum():5d}  min {v.min():7d}  median {int(np.median(v)):7d}  p90 {int(np.percentile(v, 90)):7d}  max {v.max():7d}")
            if k == 1:
                names = {0: "tests done", 1: "list atomic returned (wave 0)", 2: "list stores issued", 3: "after the coarse-bound barriers", 4: "rows x / y arrived", 5: "position-order stores issued"}
                for j in range(8):
                    col = fine[m][:, j]
                    ok = col > 1000000
                    if ok.any():
                        v = 10 * (col[ok] - t0)
                        print(f"   fine {j} ({names.get(j, '')}): blocks {ok.sum():5d}  min {v.min():7d}  median {int(np.median(v)):7d}  p90 {int(np.percentile(v, 90)):7d}  max {v.max():7d}")
d.close()
