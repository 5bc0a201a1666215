"""The row-sharded NJ loop across PROCESSES on one GPU: every rank is its own process with its own HIP context, the
other ranks' matrix rows and windows are mapped through hipIpc handles, records travel through the mailboxes and rows
x / y are pulled from their owner's memory -- the same code path as one process per GPU over xGMI, minus the link.
(RCCL cannot be used here: it refuses two ranks on one device.)  Merge logs must equal the oracle's on every rank."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import _util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(tmp_path, world, n, seed, source="matrix", timeout=240, extra_env=None):
    procs = []
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.update(extra_env or {})
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, "-m", "tests._njs_worker", str(r), str(world), str(n), str(seed),
                                       str(tmp_path / ("r%d.npz" % r)), source], cwd=ROOT, env=env, stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        blobs = [p.stdout.readline().strip() for p in procs]
        assert all(len(b) == 384 for b in blobs), [p.stderr.read()[-2000:] for p, b in zip(procs, blobs) if len(b) != 384]
        for p in procs:
            p.stdin.write("\n".join(blobs) + "\n")
            p.stdin.flush()
        outs = [p.communicate(timeout=timeout) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "done" in so, se[-3000:]
    return [np.load(tmp_path / ("r%d.npz" % r)) for r in range(world)]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n", [(2, 300), (3, 777), (4, 130)])
def test_process_ranks_on_one_gpu_match_oracle(tmp_path, orc, world, n):
    seed = 100 * world + n
    res = _run_ranks(tmp_path, world, n, seed)
    D = _util.random_additive_matrix(np.random.default_rng(seed), n, zero_frac=0.3)
    ref = orc.nj_run(np.tril(D, -1))
    for r, got in enumerate(res):
        assert str(got["plan"]) == "mailbox" and int(got["collectives"]) == 0
        assert int(got["iters"]) == n - 2
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(got[key], ref[key]), (r, key)
        assert float(got["last_d"]) == ref["last_d"]


@pytest.mark.timeout(600)
def test_process_ranks_detect_a_corrupted_pull(tmp_path):
    """dpr_ctx_set_debug_fault(40, 1) (through the worker's DPR_TEST_NJS_FAULT=40,1): rank 1 of 3 corrupts one element of a row it pulled at iteration 40.  Its replicated row sums
    then differ from the other ranks'; every rank's record carries the bits of the row sum it derived (NjsRec::ux), so
    POST(41) sees the difference on EVERY rank and all three runs end with DPR_ERR_COMM (-5) instead of three merge logs
    of which one is silently wrong."""
    res = _run_ranks(tmp_path, 3, 400, 7, extra_env={"DPR_TEST_NJS_FAULT": "40,1"})
    for got in res:
        v = str(got["verdict"])
        assert v.startswith("code -5") and "row sums differ after 41 iterations" in v, v


@pytest.mark.timeout(600)
def test_process_ranks_msa_source(tmp_path):
    """aligned input: every rank computes the distance rows it owns; all ranks end with the single-GPU merge log"""
    import dipper_amd
    from dipper_amd import capi
    world, n, seed = 2, 500, 42
    res = _run_ranks(tmp_path, world, n, seed, source="msa")
    seqs = _util.synth_alignment(np.random.default_rng(seed), n, 800, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    one = dipper_amd.Dipper(0)
    try:
        one.set_nj_mode(0)
        one.set_msa(capi.pack4_many(seqs), 800)
        one.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        ref = one.nj_run()
    finally:
        one.close()
    for got in res:
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(got[key], ref[key]), key
        assert float(got["last_d"]) == ref["last_d"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n", [(2, 2500), (3, 1300), (3, 3300)])
def test_process_ranks_row_sharded_pruned_match_oracle(tmp_path, orc, world, n):
    """the row-sharded exact PRUNED NJ (njr.hip) with one process per rank on one GPU: each process holds its chunks of the
    position-space rows only; block records and the winner's two columns travel through the mailboxes of the hipIpc-mapped
    windows (no collective), the epoch rebuilds pull source rows out of the other processes' buffers; every rank ends with the
    oracle's merge log (src/neighborJoining.cu:197-249), epochs rebuilt on the way, the run resumed once."""
    seed = 1000 * world + n
    res = _run_ranks(tmp_path, world, n, seed, extra_env={"DPR_TEST_NJ_ROWS_PRUNED": "1", "DPR_NJ_EPOCH_MIN": "300"})
    D = _util.random_additive_matrix(np.random.default_rng(seed), n, zero_frac=0.3)
    ref = orc.nj_run(np.tril(D, -1), threads=8)
    for r, got in enumerate(res):
        assert int(got["collectives"]) == 0 and int(got["launches"]) > 0
        assert int(got["iters"]) == n - 2
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(got[key], ref[key]), (r, key)
        assert float(got["last_d"]) == ref["last_d"]


@pytest.mark.timeout(600)
def test_process_ranks_row_sharded_pruned_msa_source(tmp_path):
    """aligned input, 3 000 tips (natural epochs): distance rows computed by their owners, sharded build of epoch 0 out of the
    tip-order rows of both processes; both ranks end with the single-GPU pruned merge log"""
    import dipper_amd
    from dipper_amd import capi
    world, n, seed = 2, 3000, 43
    res = _run_ranks(tmp_path, world, n, seed, source="msa", extra_env={"DPR_TEST_NJ_ROWS_PRUNED": "1"})
    seqs = _util.synth_alignment(np.random.default_rng(seed), n, 800, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    one = dipper_amd.Dipper(0)
    try:
        one.set_nj_mode(1)
        one.set_msa(capi.pack4_many(seqs), 800)
        one.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        ref = one.nj_run()
    finally:
        one.close()
    for got in res:
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(got[key], ref[key]), key
        assert float(got["last_d"]) == ref["last_d"]


@pytest.mark.timeout(300)
def test_large_export_is_refused_on_an_old_runtime_instead_of_hanging():
    """A process that imports torch runs the library on the wheel's HIP runtime; 7.0 never returns from hipIpcOpenMemHandle
    for an allocation of 2^31 .. 2^32 bytes (profiles/r3/ipc_runtime_probe.txt).  The library must refuse to describe such a
    matrix to the peers there (dpr_peer_export fails with a message; dpr_dist_matrix over RCCL falls back to the legacy
    loop) -- and must still export it on the system runtime."""
    code = r"""
import ctypes, sys
if sys.argv[1] == "torch":
    import torch
import dipper_amd
from dipper_amd import capi
capi.load_library()
ver = ctypes.c_int(0)
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        ctypes.CDLL(line.split()[-1]).hipRuntimeGetVersion(ctypes.byref(ver))
        break
d = dipper_amd.Dipper(0)
d.set_nj_mode(0)
d.comm_init_local(0, 2)
try:
    blob = d.peer_export(24000)          # 12000 rows x 24000 x 8 = 2.3 GB: bit 31 set
    print("RESULT", ver.value, "exported", len(blob))
except capi.DipperError as e:
    print("RESULT", ver.value, "refused", str(e))
d.close()
"""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for mode in ("plain", "torch"):
        r = subprocess.run([sys.executable, "-c", code, mode], cwd=ROOT, env=env, capture_output=True, text=True, timeout=200)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        _, ver, verdict, rest = line[0].split(" ", 3)
        if int(ver) >= 70200000:
            assert verdict == "exported" and rest == "192", line
        else:
            assert verdict == "refused" and "older than 7.2" in rest, line
