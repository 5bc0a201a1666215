"""CPU tests (no GPU) of the rank rendezvous of `dipper --gpus G` / dpr_comm_init_shared: the shared host region's barrier, its
512-byte-per-rank gather and the failure word, driven by several PROCESSES through the library's host-only entry points
(dpr_shared_barrier / dpr_shared_gather / dpr_shared_abort: include/dipper_hip.h) -- the same code the ranks run before and between
their device collectives -- and the command's launcher when its ranks cannot start (no GPU here: every rank fails in dpr_create)."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import time

import pytest

from dipper_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dipper_amd", "bin", "dipper")


def _worker(rank, world, name, rounds, q, die_at=-1):
    reg = capi.SharedRegion(name)
    sense = C.c_uint32(0)
    ok = True
    try:
        for it in range(rounds):
            if it == die_at:
                os._exit(3)                  # a rank that dies without a word (the launcher's job to notice)
            mine = bytes([rank, it % 251]) * 16 + rank.to_bytes(4, "little")
            got = reg.gather(rank, world, sense, mine, timeout_ms=20000)
            ok = ok and all(g == bytes([r, it % 251]) * 16 + r.to_bytes(4, "little") for r, g in enumerate(got))
            reg.barrier(world, sense, timeout_ms=20000)
        q.put((rank, "ok" if ok else "mismatch"))
    except capi.DipperError as e:
        q.put((rank, "error %d: %s" % (e.code, e)))


@pytest.mark.parametrize("world", [2, 3, 5])
def test_barrier_and_gather_between_processes(world):
    name = "dpr_test_region_%d_%d" % (os.getpid(), world)
    reg = capi.SharedRegion(name, create=True)
    try:
        q = mp.Queue()
        ps = [mp.Process(target=_worker, args=(r, world, name, 300, q)) for r in range(world)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=120) for _ in ps)
        for p in ps:
            p.join(30)
        assert res == [(r, "ok") for r in range(world)]
        assert not reg.failed()
    finally:
        reg.unlink()


def test_failure_word_ends_every_wait():
    """a rank dies at round 7; the launcher (here: the test) raises the failure word; the survivors leave their barrier with
    DPR_ERR_COMM (-5) at once instead of waiting for the time limit"""
    name = "dpr_test_region_fail_%d" % os.getpid()
    reg = capi.SharedRegion(name, create=True)
    try:
        q = mp.Queue()
        ps = [mp.Process(target=_worker, args=(r, 3, name, 1000, q, 7 if r == 1 else -1)) for r in range(3)]
        for p in ps:
            p.start()
        ps[1].join(60)
        assert ps[1].exitcode == 3
        t0 = time.time()
        reg.abort()
        res = sorted(q.get(timeout=60) for _ in range(2))
        assert time.time() - t0 < 10
        for p in ps:
            p.join(30)
        assert [r for r, _ in res] == [0, 2] and all(m.startswith("error -5") and "another rank failed" in m for _, m in res), res
        assert reg.failed()
    finally:
        reg.unlink()


def test_a_wait_is_bounded():
    """one rank alone in a barrier of two: DPR_ERR_COMM after the time limit, and the failure word is raised for the others"""
    name = "dpr_test_region_timeout_%d" % os.getpid()
    reg = capi.SharedRegion(name, create=True)
    try:
        sense = C.c_uint32(0)
        t0 = time.time()
        with pytest.raises(capi.DipperError) as ei:
            reg.barrier(2, sense, timeout_ms=300)
        assert ei.value.code == -5 and "timed out" in str(ei.value) and 0.25 < time.time() - t0 < 5
        assert reg.failed()
    finally:
        reg.unlink()


def test_launcher_reports_failing_ranks_and_exits_1(tmp_path):
    """`dipper --gpus 3` on a host without a GPU: the input is read, three ranks are started, each fails in dpr_create, the launcher
    names the first and exits 1 -- it never touches the GPU itself and does not hang"""
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build()
    fa = tmp_path / "a.fa"
    fa.write_text("".join(">t%d\nACGTACGTAC%s\n" % (i, "ACGT"[i % 4]) for i in range(6)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    t0 = time.time()
    r = subprocess.run([BIN, "-i", "m", "-I", str(fa), "-O", str(tmp_path / "o.nwk"), "-m", "2", "--gpus", "3"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 1 and time.time() - t0 < 60
    assert "Starting 3 ranks (devices 0 1 2)" in r.stderr and "ERROR: rank" in r.stderr and "failed" in r.stderr, r.stderr[-1500:]
    r = subprocess.run([BIN, "-i", "m", "-I", str(fa), "-O", str(tmp_path / "o.nwk"), "--world", "2"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 1 and "--world needs --rank" in r.stderr
    r = subprocess.run([BIN, "-i", "m", "-I", str(fa), "-O", str(tmp_path / "o.nwk"), "--transport", "smoke"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 1 and "--transport: auto, rccl or ipc" in r.stderr
