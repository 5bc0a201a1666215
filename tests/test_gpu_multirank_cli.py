"""`dipper --gpus G`: several GPUs through the product's own boundary, rehearsed on ONE GPU.

The command forks one rank process per entry of --devices once the input is read (dipper_amd/host/io.cpp: startRanks); the ranks
join through the shared region (dpr_comm_init_shared).  With `--devices 0,0[,0]` the ranks share the GPU of the box, so the
transport is the device windows over hipIpc (RCCL refuses two ranks on one device) -- everything else is the code that runs with
one rank per GPU: NJ with replicated / unit-sharded / row-sharded matrices, placement and --add with the distance rows of a
batch sharded + one all-gather per batch (src/placement_close_k.cu:756-851,858-990), divide-and-conquer with query shares and
clusters dealt to the ranks + all-reduces (src/divide_and_conquer/placement_close_k.cu:731-1535).  The bar: the Newick file is
BYTE-IDENTICAL to the one-rank run of the same command (which the other GPU suites compare with the oracle)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
GEN = os.path.join(ROOT, "tools", "bin", "gen_synth")


def run(*args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([BIN, *args], capture_output=True, text=True, env=e, timeout=timeout)


def gen(path, tips, sites, seed=1, extra=()):
    r = subprocess.run([GEN, "--tips", str(tips), "--sites", str(sites), "--seed", str(seed), "--mean-bl", "0.004", "--lo", "0.0004",
                        "--hi", "0.04", "--fasta", str(path), *extra], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def both(tmp_path, args, devices, env=None, tag="x"):
    """the command with one rank and with the ranks of `devices`: (one-rank text, multi-rank text, multi-rank stderr)"""
    o1, oG = tmp_path / (tag + "_1.nwk"), tmp_path / (tag + "_G.nwk")
    r1 = run(*args, "-O", str(o1), env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    rG = run(*args, "-O", str(oG), "--devices", devices, env=env)
    assert rG.returncode == 0, rG.stderr[-3000:]
    n = len(devices.split(","))
    assert f"Starting {n} ranks" in rG.stderr and f"Ranks: {n} (transport ipc" in rG.stderr, rG.stderr[-1500:]
    return o1.read_text(), oG.read_text(), rG.stderr


def collectives(stderr):
    line = [l for l in stderr.splitlines() if l.startswith("Ranks: ")][-1]
    return int(line.split("transport ipc, ")[1].split(" ")[0])


@pytest.fixture(scope="module")
def aligned(tmp_path_factory):
    p = tmp_path_factory.mktemp("mr") / "aln.fa"
    gen(p, 2600, 700)
    return p


@pytest.mark.timeout(900)
@pytest.mark.parametrize("devices,env,expect", [
    ("0,0", {}, "replicas"),                                               # default below 65 536 tips: every rank the single-GPU plan
    ("0,0", {"DPR_NJ_MULTI": "shard"}, "unit tests and scans sharded"),    # one all-gather of block records per iteration
    ("0,0,0", {"DPR_NJ_MULTI": "rows"}, "row-sharded pruned NJ"),          # njr.hip: rows of the position-space matrix sharded
    ("0,0", {"DPR_NJ_MODE": "stream"}, "streaming, rows sharded"),         # njs.hip: north_star's row blocks, mailbox exchange
    ("0,0,0", {"DPR_NJ_MODE": "stream", "DPR_NJ_EXCHANGE": "legacy"}, "streaming, rows sharded"),
])
def test_nj_command_with_ranks_equals_one_rank(tmp_path, aligned, devices, env, expect):
    """-m 2 (src/neighborJoining.cu:197-249 over src/tree_generation.cu:576-592) with 2 - 3 ranks in every multi-rank NJ plan"""
    a, b, err = both(tmp_path, ["-i", "m", "-I", str(aligned), "-m", "2", "-d", "2"], devices, env=env)
    assert a == b and a.count(",") == 2599
    assert expect in err, err[-1500:]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,devices", [("m", "0,0"), ("m", "0,0,0"), ("r", "0,0")])
def test_placement_command_with_ranks_equals_one_rank(tmp_path, aligned, kind, devices):
    """-m 1: the distance rows of every batch computed in shares and all-gathered (ctx_place.hip: place_range), the tree kernels
    replicated; the all-gathers really ran (the rank summary counts them)"""
    extra = ["-d", "2"] if kind == "m" else []
    a, b, err = both(tmp_path, ["-i", kind, "-I", str(aligned), "-m", "1", *extra], devices)
    assert a == b and a.count(",") == 2599
    assert collectives(err) >= 2600 // 1024


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,devices", [("m", "0,0,0"), ("r", "0,0")])
def test_add_command_with_ranks_equals_one_rank(tmp_path, kind, devices):
    """--add (src/placement_close_k.cu:858-990): 700 queries onto a 1 500-tip backbone built by the command itself"""
    fa_all, fa_b = tmp_path / "all.fa", tmp_path / "b.fa"
    gen(fa_all, 2200, 600, seed=5)
    # the backbone: the first 1 500 records
    recs = fa_all.read_text().split(">")[1:]
    fa_b.write_text("".join(">" + r for r in recs[:1500]))
    extra = ["-d", "2"] if kind == "m" else []
    bb = tmp_path / "bb.nwk"
    assert run("-i", kind, "-I", str(fa_b), "-O", str(bb), "-m", "1", *extra).returncode == 0
    a, b, err = both(tmp_path, ["-i", kind, "-I", str(fa_all), "--add", "-t", str(bb), *extra], devices)
    assert a == b and a.count(",") == 2199
    assert collectives(err) >= 1


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind,devices", [("m", "0,0"), ("m", "0,0,0"), ("r", "0,0,0")])
def test_dc_command_with_ranks_equals_one_rank(tmp_path, kind, devices):
    """-m 3 (src/tree_generation.cu:422-449,541-575): backbone replicated, query shares and clusters dealt to the ranks, cluster ids
    and state deltas summed over the transport (ctx_place.hip: dpr_dc_run)"""
    fa = tmp_path / "dc.fa"
    gen(fa, 6000, 500, seed=9)
    extra = ["-d", "2"] if kind == "m" else []
    a, b, err = both(tmp_path, ["-i", kind, "-I", str(fa), "-m", "3", *extra], devices)
    assert a == b and a.count(",") == 5999
    assert collectives(err) >= 10       # 1 all-reduce of the cluster ids + 9 of the state arrays (+ the backbone's batches)


@pytest.mark.timeout(300)
def test_a_failing_rank_ends_the_command_instead_of_hanging(tmp_path, aligned):
    """rank 1 names a device that does not exist: it fails in dpr_create while rank 0 waits in dpr_comm_init_shared; the launcher
    raises the region's failure word, rank 0 leaves its wait with an error, the command exits 1 -- within seconds"""
    import time
    t0 = time.time()
    r = run("-i", "m", "-I", str(aligned), "-m", "1", "-O", str(tmp_path / "o.nwk"), "--devices", "0,97", timeout=120)
    assert r.returncode == 1 and time.time() - t0 < 60
    assert "rank 1" in r.stderr and "failed" in r.stderr, r.stderr[-1500:]


@pytest.mark.timeout(600)
def test_ranks_started_from_outside_meet_in_a_named_region(tmp_path, aligned):
    """--rank / --world / --rendezvous: two processes started by the test (as mpirun or srun would) find each other through a
    POSIX shared memory object; rank 0 writes the tree of the one-rank run"""
    o1 = tmp_path / "one.nwk"
    assert run("-i", "m", "-I", str(aligned), "-m", "1", "-d", "2", "-O", str(o1)).returncode == 0
    name = "dipper_test_%d" % os.getpid()
    outs = [tmp_path / ("ext%d.nwk" % r) for r in range(2)]
    ps = [subprocess.Popen([BIN, "-i", "m", "-I", str(aligned), "-m", "1", "-d", "2", "-O", str(outs[r]), "--device", "0", "--devices", "0,0",
                            "--rank", str(r), "--world", "2", "--rendezvous", name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for r in range(2)]
    try:
        res = [p.communicate(timeout=500) for p in ps]
    finally:
        for p in ps:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink("/dev/shm/" + name)
        except OSError:
            pass
    assert all(p.returncode == 0 for p in ps), [e[-1500:] for _, e in res]
    assert outs[0].read_text() == o1.read_text()
    assert not outs[1].exists()               # (rank 1 writes nothing)
    assert "Ranks: 2 (transport ipc" in res[0][1]


@pytest.mark.timeout(600)
def test_ranks_fall_back_to_the_windows_when_rccl_does_not_come_up(tmp_path, aligned):
    """`--transport auto` with one rank per GPU means RCCL; when RCCL does not come up on every rank (here: DPR_TEST_COMM_TRY_RCCL=1 makes
    two ranks that share GPU 0 try it, and RCCL refuses a duplicate device) the ranks fall back TOGETHER to the device windows and the
    run goes on: same Newick, transport ipc in the closing line, a note on stderr.  An explicit `--transport rccl` is an error instead."""
    o1, o2 = tmp_path / "one.nwk", tmp_path / "two.nwk"
    assert run("-i", "m", "-I", str(aligned), "-m", "1", "-d", "2", "-O", str(o1)).returncode == 0
    r = run("-i", "m", "-I", str(aligned), "-m", "1", "-d", "2", "-O", str(o2), "--devices", "0,0", env={"DPR_TEST_COMM_TRY_RCCL": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RCCL did not come up on every rank" in r.stderr and "Ranks: 2 (transport ipc" in r.stderr, r.stderr[-1500:]
    assert o2.read_text() == o1.read_text()
    r = run("-i", "m", "-I", str(aligned), "-m", "1", "-d", "2", "-O", str(o2), "--devices", "0,0", "--transport", "rccl")
    assert r.returncode == 1 and "RCCL refuses two ranks on one device" in r.stderr


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["2", "1"], ids=["nj", "placement"])
@pytest.mark.parametrize("tips,devices", [(4, "0,0,0,0,0"), (7, "0,0,0"), (1501, "0,0,0,0,0")],
                         ids=["4_tips_5_ranks", "7_tips_3_ranks", "1501_tips_5_ranks"])
def test_more_ranks_than_work_and_uneven_shares(tmp_path, tips, devices, mode):
    """five process ranks (the most a one-GPU box admits beside the launcher), fewer tips than ranks, shares that do not divide:
    empty shares and ragged all-gather segments must leave the Newick file as the one-rank run writes it"""
    p = tmp_path / "small.fa"
    gen(p, tips, 300, seed=tips)
    a, b, _ = both(tmp_path, ["-i", "m", "-I", str(p), "-m", mode, "-d", "2"], devices)
    assert a == b and a.count(",") == tips - 1


@pytest.mark.timeout(600)
def test_exact_placement_command_with_ranks_equals_one_rank(tmp_path, aligned):
    """-m 1 -p 0 (src/placement.cu): the exact mode has no sharded part -- every rank runs the whole placement, rank 0 writes the
    tree; the file must be the one-rank file (the per-batch read-back that sizes the small subtrees runs on every rank's own stream)"""
    a, b, err = both(tmp_path, ["-i", "m", "-I", str(aligned), "-m", "1", "-p", "0", "-d", "2"], "0,0")
    assert a == b and a.count(",") == 2599
    assert "exact placement mode" in err
