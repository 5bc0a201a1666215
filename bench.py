#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json):
tips/sec for packed aligned tips -> JC distances -> conventional NJ -> merge log at N = 30k on
MI355X, plus the Q-argmin HBM roofline and a CPU NJ baseline timed on the host cores.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident
in HBM (bit-plane sequences): all-pairs JC69 distance matrix + row sums + all N-2 NJ iterations.
Run as `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches it under
torch.distributed.run (one rank per GPU, RCCL).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """Threads the CPU baselines may use: the smallest of os.cpu_count(), the affinity mask and the cgroup CPU
    quota (a GPU box exposes 256 logical CPUs but grants 16; oversubscribed OpenMP teams run many times slower)."""
    c = os.cpu_count() or 1
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            c = min(c, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, c)


def make_input(n, L, seed):
    """Seeded stand-in for `iqtree2 --alisim` (scripts/alisim.sh:14): Yule-Harding tree, JC69,
    branch lengths exponential(2e-5) clipped to [2e-6, 2e-4], no indels."""
    from tests import _util
    rng = np.random.default_rng(seed)
    return _util.synth_alignment(rng, n, L, mean_bl=2e-5, lo=2e-6, hi=2e-4)


def pmc_traffic(n, world):
    """HBM bytes per scan launch from the committed rocprofv3 --pmc passes of
    `bench.py --probe-only` (profiles/pmc_scan.json; FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md, WRITE_SIZE as is).  None when no matching measurement is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_scan.json")) as f:
            rec = json.load(f)
        if rec.get("n_active") == n and rec.get("n_gpus") == world:
            return rec["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(dip, n, budget_s=20.0):
    """CPU NJ (the oracle's restatement of the reference's arithmetic, OpenMP over row bands) on a
    bounded sample: the first k iterations at full size on the same matrix, extrapolated to the
    whole run by the sum of n^2.  Distances are NOT included (the matrix is copied from the GPU)."""
    import psutil
    from tests import _orc
    orc = _orc.load()
    cores = host_cores()
    log(f"[cpu_baseline] oracle NJ on {cores} host threads ...")
    need = n * n * 8 * 1.15
    avail = psutil.virtual_memory().available
    ns = n
    if need > 0.5 * avail:
        ns = int((0.5 * avail / 9.2) ** 0.5)
        log(f"[cpu_baseline] host memory {avail/2**30:.0f} GiB: sampling the leading {ns} tips")
    D = np.zeros((ns, ns), dtype=np.float64)
    for i in range(ns):
        D[i, :] = dip.matrix_row(i)[:ns]
    D = np.tril(D, -1)
    # calibrate: 2 iterations, then as many as fit the budget
    import ctypes as C
    from tests._orc import _p, c_f64p, c_i32p
    k_max = 64
    mx = np.zeros(k_max, np.int32); my = np.zeros(k_max, np.int32)
    bx = np.zeros(k_max); by = np.zeros(k_max)
    last = C.c_double()

    def run(k):
        Dc = D.copy()
        t0 = time.perf_counter()
        orc.lib.orc_nj_run(_p(Dc, c_f64p), ns, ns, cores, k, _p(mx, c_i32p), _p(my, c_i32p),
                           _p(bx, c_f64p), _p(by, c_f64p), C.byref(last), None)
        return time.perf_counter() - t0

    t0 = run(0)            # mirror + row sums only
    t2 = run(2)
    per_it = max((t2 - t0) / 2, 1e-6)
    k = int(max(2, min(k_max, (budget_s - t2) / per_it)))
    tk = run(k) if k > 2 else t2
    per_it = (tk - t0) / k
    s_sample = sum(float(ns - i) ** 2 for i in range(k))
    s_full = sum(float(m) ** 2 for m in range(3, n + 1))
    t_full = t0 * (n / ns) ** 2 + per_it * k * s_full / s_sample
    return {
        "value": n / t_full, "unit": "tips/s", "cores": cores, "kind": "port",
        "sample": f"oracle NJ (reference arithmetic, OpenMP {cores} threads) on the same matrix "
                  f"(leading {ns} tips): init + first {k} iterations timed ({tk:.1f} s), "
                  f"extrapolated to all {n-2} iterations of N={n} by sum(n^2); distance stage excluded",
    }


def cpu_baseline_rapidnj(dip, n, budget_s=20.0):
    """Stronger CPU baseline next to the oracle's: a from-scratch RapidNJ-style exact NJ (sorted rows +
    q_min pruning, OpenMP; oracle/rapidnj_baseline.c -- north_star names RapidNJ, which is not installed and
    cannot be fetched).  Whole NJ run on the GPU's distance matrix when it fits the time budget (calibrated on
    the leading 6 000 tips), else on the largest leading block that does, extrapolated with the measured
    exponent.  Distances excluded, as for the oracle baseline."""
    import psutil
    from tests import _orc
    orc = _orc.load()
    cores = host_cores()
    log(f"[cpu_baseline_rapidnj] RapidNJ-style NJ on {cores} host threads ...")

    def block(m):
        D = np.zeros((m, m), dtype=np.float64)
        for i in range(m):
            D[i, :] = dip.matrix_row(i)[:m]
        return D

    def run(D):
        t0 = time.perf_counter()
        r = orc.rapidnj_run(D, threads=cores)
        dt = time.perf_counter() - t0
        assert r["joins"] == D.shape[0] - 2
        log(f"[cpu_baseline_rapidnj] {D.shape[0]} tips: {dt:.2f} s")
        return dt

    m0 = min(n, 3000)
    m1 = min(n, 6000)
    D1 = block(m1)
    t0 = run(D1[:m0, :m0])
    t1 = run(D1) if m1 > m0 else t0
    expo = max(1.5, min(3.0, np.log(max(t1, 1e-3) / max(t0, 1e-3)) / np.log(m1 / m0))) if m1 > m0 else 2.0
    est_full = t1 * (n / m1) ** expo
    avail = psutil.virtual_memory().available
    m = n
    if est_full > budget_s:
        m = int(m1 * (budget_s / max(t1, 1e-3)) ** (1.0 / expo))
    m = min(m, int((0.4 * avail / 14.0) ** 0.5))       # matrix copy + working copy + sorted rows
    m = max(m1, min(m, n))
    if m > m1:
        del D1
        tm = run(block(m))
    else:
        tm = t1
    t_full = tm * (n / m) ** expo
    return {"value": n / t_full, "unit": "tips/s", "cores": cores, "kind": "rapidnj-style reimplementation (not the oracle)",
            "sample": f"exact NJ with RapidNJ's sorted-row search on the same matrix, leading {m} of {n} tips in {tm:.1f} s "
                      + ("(whole run)" if m == n else f"extrapolated by (N/m)^{expo:.2f} (exponent measured between {m0} and {m1} tips)")
                      + "; distance stage excluded"}


def e2e_cli(seqs, n, L):
    """BASELINE.json's literal headline: wall time of the whole `dipper` command (FASTA parse, pack, H2D,
    distances, NJ, Newick write) on the same synthetic alignment, written to a scratch FASTA first.
    Reported next to `value` (which by contract excludes host I/O), never as `value`."""
    import subprocess
    import tempfile
    from tests import _util
    exe = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
    if not os.path.exists(exe):
        return {"error": "dipper_amd/bin/dipper not built"}
    with tempfile.TemporaryDirectory() as tmp:
        fa, out = os.path.join(tmp, "in.fa"), os.path.join(tmp, "out.nwk")
        _util.write_fasta(fa, ["T%d" % (i + 1) for i in range(n)], seqs, width=0)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-i", "m", "-I", fa, "-O", out, "-m", "2", "-d", "2"], capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            best = dt if best is None else min(best, dt)
        return {"metric": "tips/sec FASTA -> Newick, whole CLI run", "wall_s": best, "tips_per_s": n / best,
                "fasta_bytes": os.path.getsize(fa), "newick_bytes": os.path.getsize(out),
                "command": "dipper -i m -I in.fa -O out.nwk -m 2 -d 2", "runs": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tips", type=int, default=30000)
    ap.add_argument("--sites", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--probe-reps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end CLI timing (FASTA -> Newick)")
    ap.add_argument("--probe-only", action="store_true",
                    help="skip the timed steps; only build the matrix and run the roofline probe "
                         "(used for the rocprofv3 --pmc passes)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: everything else that writes to file descriptor 1 (RCCL prints
    # a version banner there when a communicator is created) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using WORLD_SIZE")

    import torch
    import dipper_amd
    from dipper_amd import capi

    dist = None
    # DPR_BENCH_CHECK=1 under `torch.distributed.run --nproc-per-node 1` rehearses the multi-GPU self-check on one GPU
    force_check = os.environ.get("DPR_BENCH_CHECK") == "1" and "RANK" in os.environ
    if world > 1 or force_check:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n, L = args.tips, args.sites
    t0 = time.perf_counter()
    seqs = make_input(n, L, args.seed)
    packed = capi.pack4_many(seqs)
    want_e2e = rank == 0 and world == 1 and not args.no_e2e and not args.probe_only
    if not want_e2e:
        del seqs
    log(f"[bench r{rank}] synthetic input {n} x {L} generated+packed in {time.perf_counter()-t0:.1f}s")

    dip = dipper_amd.Dipper(local_rank)
    comm_note = None
    if world > 1:
        # the library's own RCCL communicator; if any rank cannot create it, every rank falls back to the
        # single-GPU plan on its own GPU (same result, no exchange) and the bench line says so
        ok, err = 1, ""
        try:
            uid = [dip.comm_unique_id() if rank == 0 else None]
        except Exception as e:
            uid, ok, err = [None], 0, repr(e)
        dist.broadcast_object_list(uid, src=0)
        if ok and uid[0] is not None:
            try:
                dip.comm_init(rank, world, uid[0])
            except Exception as e:
                ok, err = 0, repr(e)
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            comm_note = "replicated: the library's RCCL communicator could not be created on every rank (%s)" % (err or "another rank failed")
            log(f"[bench r{rank}] {comm_note}")
            dip.close()
            dip = dipper_amd.Dipper(local_rank)
    dip.set_msa(packed, L)          # H2D + bit-plane conversion: inputs now resident in HBM
    if rank == 0:
        log(f"[bench] device: {dip.device_name()}")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        dip.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        res = dip.nj_run()
        assert res["iters"] == n - 2
        return res

    phase = []
    prune = None
    last_res = None
    if args.probe_only:
        args.steps = args.warmup = 0
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last_res = step()
        phase.append(dip.timing())
        try:
            sc, full = dip.prune_stats()
            prune = {"units_scanned": sc, "units_per_full_scan": full, "iterations": n - 2,
                     "scanned_fraction_of_full_scans": sc / (full * (n - 2.0))}
        except Exception:
            prune = None
    barrier()
    dt = time.perf_counter() - t0
    sharded_timed = world > 1 and dip.nj_is_unit_sharded()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / max(args.steps, 1) * 1e3 if args.steps else float("nan")

    # ---- several GPUs: the ranks must hold the same merge log, and it must be the one a single GPU produces ----
    mgpu_check = None
    run_sharded_check = False
    if (world > 1 or force_check) and last_res is not None:
        import hashlib
        h = hashlib.sha256()
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            h.update(np.ascontiguousarray(last_res[key]).tobytes())
        digest = int.from_bytes(h.digest()[:7], "little")
        mine = torch.tensor([digest], dtype=torch.int64, device="cuda")
        allh = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allh, mine)
        agree = all(int(t.item()) == digest for t in allh)
        mgpu_check = {"ranks_agree": bool(agree)}
        run_sharded_check = comm_note is None and not dip.nj_is_unit_sharded()   # (also in the one-rank rehearsal)
        if rank == 0:
            try:   # untimed replay of the same step on this rank's GPU alone (no communicator)
                solo = dipper_amd.Dipper(local_rank)
                solo.set_msa(packed, L)
                solo.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
                ref = solo.nj_run()
                solo.close()
                mgpu_check["matches_single_gpu"] = bool(
                    all(np.array_equal(ref[k], last_res[k]) for k in ("merge_x", "merge_y", "bl_x", "bl_y")))
            except Exception as e:
                mgpu_check["matches_single_gpu"] = None
                mgpu_check["error"] = repr(e)
        dist.barrier()

    # ---- roofline of the dominant kernel: Q-argmin scan at n = N on a fresh matrix ----------------
    dip.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    _, _, _, scan_ms = dip.argmin_once(reps=args.probe_reps)
    rows_local = capi.load_library().dpr_shard_rows(n, rank, world)
    # pruned NJ on several GPUs keeps the whole matrix on every rank (the ranks share the unit tests and scans of
    # an iteration), so the probe streams the whole triangle; the streaming algorithm is row-sharded
    replicated = prune is not None and world > 1 and comm_note is None
    sharded = replicated and sharded_timed
    whole = world == 1 or replicated or comm_note is not None      # this rank holds (and the probe streams) the whole triangle
    if whole:
        rows_local = n
    alg_bytes = (4.0 * n * n if whole else 4.0 * n * n / world) + 4.0 * n   # strict lower triangle (of this rank's rows) + U once
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n, world),
                "kernel": "nj_scan_kernel<PROBE=true,...> (probe instantiation of the full Q-argmin scan)",
                "n_active": n, "algorithmic_bytes": alg_bytes, "ms": scan_ms, "rows_local": int(rows_local)}

    out = {
        "metric": "tips/sec packed aligned tips -> JC distances -> NJ merge log at N=%d" % n,
        "value": n / (ms_per_step * 1e-3),
        "unit": "tips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (seeded Yule-Harding tree, JC69, L=%d, no indels)" % L,
        "config": {"workload": "configs[1]: %d aligned tips, -d 2 (JC69), conventional NJ" % n,
                   "tips": n, "sites": L,
                   "parallelism": comm_note or (("units%d (matrix replicated, unit tests and scans shared, one all-gather per iteration)" % world) if sharded
                                                else ("replicas%d (every rank runs the single-GPU plan: N is below the unit-sharding threshold of 65536 tips, "
                                                      "where an iteration is ~20 us of dependent latency; the unit-sharded plan runs once untimed, see multi_gpu_check)" % world) if replicated
                                                else "rows%d" % world)},
        "phase_ms": {"dist": float(np.mean([p[0] for p in phase])) if phase else None,
                     "nj": float(np.mean([p[1] for p in phase])) if phase else None},
        "roofline": roofline,
        "nj_algorithm": "exact pruned scan (njp.hip)" if prune else "full streaming scan (nj.hip)",
        "prune": prune,
    }
    if mgpu_check is not None:
        out["multi_gpu_check"] = mgpu_check
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(dip, n)
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "tips/s", "cores": host_cores(), "kind": "port",
                                   "sample": f"failed: {e!r}"}
        try:
            out["cpu_baseline_rapidnj"] = cpu_baseline_rapidnj(dip, n)
        except Exception as e:
            out["cpu_baseline_rapidnj"] = {"value": None, "unit": "tips/s", "cores": host_cores(), "sample": f"failed: {e!r}"}
    if run_sharded_check:
        # The timed steps ran the single-GPU plan on every rank (N below the sharding threshold).  Last of all, with
        # the record complete: run the unit-sharded plan (one RCCL all-gather per iteration) once, untimed, and compare.
        # A watchdog ends every rank with the record printed if that path does not come back.
        import threading

        def give_up():
            out.setdefault("multi_gpu_check", {})["unit_sharded_plan"] = {"error": "no result within 300 s"}
            if rank == 0:
                os.write(json_fd, (json.dumps(out) + "\n").encode())
            os._exit(0)

        dog = threading.Timer(300.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            capi.set_nj_multi_plan(1)
            barrier()
            ts = time.perf_counter()
            dip.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
            sh = dip.nj_run()
            barrier()
            sh_ms = (time.perf_counter() - ts) * 1e3
            same = all(np.array_equal(sh[k], last_res[k]) for k in ("merge_x", "merge_y", "bl_x", "bl_y"))
            okt = torch.tensor([1 if same else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            out["multi_gpu_check"]["unit_sharded_plan"] = {"matches": bool(int(okt.item())), "ms_per_step": sh_ms,
                                                           "nj_ms": dip.timing()[1]}
        except Exception as e:
            out["multi_gpu_check"]["unit_sharded_plan"] = {"error": repr(e)}
        finally:
            dog.cancel()
            capi.set_nj_multi_plan(0)
    dip.close()
    if want_e2e:
        try:
            out["e2e_cli"] = e2e_cli(seqs, n, L)
        except Exception as e:  # never take the bench line down
            out["e2e_cli"] = {"error": repr(e)}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
