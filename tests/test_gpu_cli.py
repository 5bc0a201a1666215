"""GPU tests of the `dipper` command line end to end (BASELINE configs[0] and a small configs[1])."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests import _util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run(*args):
    return subprocess.run([BIN, *args], capture_output=True, text=True)


def test_config0_phylip_to_newick(tmp_path, orc):
    """configs[0]: PHYLIP matrix -> conventional NJ -> Newick.  The matrix is the patristic matrix
    of the reference's dataset/t2.backbone.nwk (sub-sampled), since dataset/t2.phy is missing."""
    names_all = [nm for nm in np.load(os.path.join(GOLD, "t2_ref_tree.npz"))["name"] if not nm.startswith("node")]
    names = names_all[:300]
    nwk = open(os.path.join(GOLD, "t2.backbone.nwk")).readline()
    D = _util.patristic(nwk, names)
    phy = tmp_path / "t2_300.phy"
    _util.write_phylip_lower(str(phy), names, D)
    out = tmp_path / "t2_300.nwk"
    r = run("-i", "d", "-I", str(phy), "-O", str(out))
    assert r.returncode == 0, r.stderr
    assert "Using conventional NJ" in r.stderr
    # oracle on the float-rounded values the reader must produce (stof, src/matrix_reader.cu:42)
    Dr = np.zeros_like(D)
    for i in range(len(names)):
        for j in range(i):
            Dr[i, j] = orc.phylip_value("%.9g" % D[i, j])
    ref = orc.nj_run(Dr)
    expect = _util.newick_from_merges(names, ref["merge_x"], ref["merge_y"], ref["bl_x"], ref["bl_y"], ref["last_d"])
    assert out.read_text() == expect
    # square matrix + space separators give the same tree (only the lower triangle is used)
    sq = tmp_path / "sq.phy"
    with open(sq, "w") as f:
        f.write(f"{len(names)}\n")
        for i, nm in enumerate(names):
            f.write(nm + " " + " ".join("%.9g" % D[i, j] for j in range(len(names))) + "\n")
    out2 = tmp_path / "sq.nwk"
    assert run("-i", "d", "-I", str(sq), "-O", str(out2)).returncode == 0
    assert out2.read_text() == expect


@pytest.mark.parametrize("gz", [False, True])
def test_msa_fasta_to_newick_uncorrected(tmp_path, orc, gz):
    """-i m -d 1 -m 2: integer counts and one fp64 division -> the Newick text is bit-identical."""
    rng = np.random.default_rng(21)
    n, L = 150, 900
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2, invalid_frac=0.02)
    names = [f"T{i+1}" for i in range(n)]
    fa = tmp_path / ("a.fa.gz" if gz else "a.fa")
    plain = tmp_path / "plain.fa"
    _util.write_fasta(str(plain), names, seqs, width=70)
    if gz:
        fa.write_bytes(gzip.compress(plain.read_bytes()))
    else:
        fa = plain
    out = tmp_path / "o.nwk"
    r = run("-i", "m", "-I", str(fa), "-O", str(out), "-m", "2", "-d", "1", "--seed", "-1")
    assert r.returncode == 0, r.stderr
    D = orc.msa_dist_lower(orc.pack4_many(seqs), L, 1)
    ref = orc.nj_run(D)
    expect = _util.newick_from_merges(names, ref["merge_x"], ref["merge_y"], ref["bl_x"], ref["bl_y"], ref["last_d"])
    assert out.read_text() == expect


def test_msa_jc_default_shuffle(tmp_path, orc):
    """-d 2 with the shuffle on: same splits as the oracle, branch lengths within 1e-6 relative
    (+-1 in the 6th printed digit)."""
    rng = np.random.default_rng(22)
    n, L = 120, 3000
    seqs = _util.synth_alignment(rng, n, L, mean_bl=2e-2, lo=5e-3, hi=1e-1)
    names = [f"S{i}" for i in range(n)]
    fa = tmp_path / "b.fa"
    _util.write_fasta(str(fa), names, seqs)
    out = tmp_path / "o.nwk"
    r = run("-i", "m", "-I", str(fa), "-O", str(out), "-m", "2", "-d", "2", "--seed", "5")
    assert r.returncode == 0, r.stderr
    D = orc.msa_dist_lower(orc.pack4_many(seqs), L, 2)
    ref = orc.nj_run(D)
    expect = _util.newick_from_merges(names, ref["merge_x"], ref["merge_y"], ref["bl_x"], ref["bl_y"], ref["last_d"])
    got = out.read_text()
    assert _util.splits(got, names) == _util.splits(expect, names)
    P1, P2 = _util.patristic(got, names), _util.patristic(expect, names)
    assert np.allclose(P1, P2, rtol=2e-5, atol=1e-9)


def _api_matrix(kind, seqs, L=None, dt=2):
    import dipper_amd
    from dipper_amd import capi
    d = dipper_amd.Dipper(0)
    try:
        if kind == "m":
            d.set_msa(capi.pack4_many(seqs), L)
            d.dist_matrix(capi.SRC_MSA, dt)
        else:
            d.set_reads(seqs)
            d.sketch(15, 1000, fetch=False)
            d.dist_matrix(capi.SRC_MASH, 0, 15)
        return d.matrix()
    finally:
        d.close()


def test_placement_modes_end_to_end(tmp_path, orc):
    """-m 1 for aligned, unaligned and matrix input: the Newick text equals the oracle's placement
    run on the same distances (src/placement_close_k.cu:646-854 + printTree :568-643)."""
    rng = np.random.default_rng(31)
    n, L = 180, 1200
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    names = [f"T{i+1}" for i in range(n)]
    fa = tmp_path / "a.fa"
    _util.write_fasta(str(fa), names, seqs)
    for kind, extra in (("m", ["-d", "2"]), ("r", [])):
        out = tmp_path / f"{kind}.nwk"
        r = run("-i", kind, "-I", str(fa), "-O", str(out), "-m", "1", "--seed", "-1", *extra)
        assert r.returncode == 0, r.stderr
        assert "k-closest placement mode" in r.stderr
        # the reference's two progress lines of a placement run (src/placement_close_k.cu:852-853)
        lines = r.stderr.splitlines()
        di = [i for i, ln in enumerate(lines) if ln.startswith("Distance Operation Time ") and ln.endswith(" ms")]
        ti = [i for i, ln in enumerate(lines) if ln.startswith("Tree Operation Time ") and ln.endswith(" ms")]
        assert len(di) == 1 and len(ti) == 1 and ti[0] == di[0] + 1, r.stderr
        M = _api_matrix(kind, seqs, L)
        st = orc.place_run(M)
        assert out.read_text() == _util.newick_from_placement(names, st["head"], st["e"], st["nxt"], st["len"], n)
    # matrix input
    D = _util.random_additive_matrix(rng, 150)
    D *= 0.9 / D.max()
    nm = [f"X{i}" for i in range(150)]
    phy = tmp_path / "d.phy"
    _util.write_phylip_lower(str(phy), nm, D)
    out = tmp_path / "d.nwk"
    assert run("-i", "d", "-I", str(phy), "-O", str(out), "-m", "1").returncode == 0
    Dr = np.zeros_like(D)
    for i in range(150):
        for j in range(i):
            Dr[i, j] = Dr[j, i] = orc.phylip_value("%.9g" % D[i, j])
    st = orc.place_run(Dr)
    assert out.read_text() == _util.newick_from_placement(nm, st["head"], st["e"], st["nxt"], st["len"], 150)


def test_exact_placement_end_to_end(tmp_path, orc):
    """-m 1 -p 0: exact placement mode (src/placement.cu); Newick text equal to the oracle's."""
    rng = np.random.default_rng(61)
    n, L = 260, 1200
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    names = [f"T{i+1}" for i in range(n)]
    fa = tmp_path / "a.fa"
    _util.write_fasta(str(fa), names, seqs)
    out = tmp_path / "x.nwk"
    r = run("-i", "m", "-I", str(fa), "-O", str(out), "-m", "1", "-p", "0", "-d", "2", "--seed", "-1")
    assert r.returncode == 0, r.stderr
    assert "exact placement mode" in r.stderr
    st = orc.place_exact_run(_api_matrix("m", seqs, L))
    assert out.read_text() == _util.newick_from_placement(names, st["head"], st["e"], st["nxt"], st["len"], n)


def test_divide_and_conquer_end_to_end(tmp_path, orc):
    """-m 3 (src/tree_generation.cu:422-449,541-575): backbone = N/20, cluster assignment, cluster
    trees, printTreeDC; Newick text equal to the oracle's sequential restatement on the same distances."""
    rng = np.random.default_rng(51)
    n, L = 1400, 1200
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    names = [f"T{i+1}" for i in range(n)]
    fa = tmp_path / "a.fa"
    _util.write_fasta(str(fa), names, seqs)
    for kind, extra, skip in (("m", ["-d", "2"], 1), ("r", [], 0)):
        out = tmp_path / f"dc_{kind}.nwk"
        r = run("-i", kind, "-I", str(fa), "-O", str(out), "-m", "3", "--seed", "-1", *extra)
        M = _api_matrix(kind, seqs, L)
        st = orc.dc_run(M, n // 20, skip_last_backbone=skip)
        if st["next_slot"] == -2:      # a cluster as large as the backbone: the reference stops, so do we
            assert r.returncode == 1 and "not fewer than the backbone size" in r.stderr
            continue
        assert r.returncode == 0, r.stderr
        assert "Using divide-and-conquer mode" in r.stderr and "Finished clustering in" in r.stderr
        assert out.read_text() == _util.newick_from_placement(names, st["head"], st["e"], st["nxt"], st["len"], n)


@pytest.mark.parametrize("kind", ["m", "r"])
def test_add_queries_to_backbone(tmp_path, orc, kind):
    """--add (src/tree_generation.cu:252-332, addQuery src/placement_close_k.cu:858-990)."""
    rng = np.random.default_rng(77)
    n, m, L = 240, 180, 1500
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    names = [f"T{i+1}" for i in range(n)]
    # shuffle the file order so that backbone tips and queries interleave
    perm = rng.permutation(n)
    back = sorted(perm[:m].tolist())
    fa_b, fa_all = tmp_path / "b.fa", tmp_path / "all.fa"
    _util.write_fasta(str(fa_b), [names[i] for i in back], [seqs[i] for i in back])
    _util.write_fasta(str(fa_all), [names[i] for i in perm], [seqs[i] for i in perm])
    bb = tmp_path / "backbone.nwk"
    extra = ["-d", "2"] if kind == "m" else []
    assert run("-i", kind, "-I", str(fa_b), "-O", str(bb), "-m", "1", "--seed", "3", *extra).returncode == 0
    out = tmp_path / "added.nwk"
    r = run("-i", kind, "-I", str(fa_all), "-O", str(out), "--add", "-t", str(bb), *extra)
    assert r.returncode == 0, r.stderr
    # oracle mirror
    st, leaf_names = _util.backbone_state(orc, bb.read_text(), n)
    assert len(leaf_names) == m
    order_names = list(leaf_names) + [names[i] for i in perm if names[i] not in set(leaf_names)]
    by_name = dict(zip(names, seqs))
    M = _api_matrix(kind, [by_name[x] for x in order_names], L)
    orc.place_init_lists(n, m, st)
    st = orc.place_run(M, first=m, state=st)
    assert out.read_text() == _util.newick_from_placement(order_names, st["head"], st["e"], st["nxt"], st["len"], n)


def test_output_distance_matrix(tmp_path, orc):
    """-o d writes the lower-triangular PHYLIP matrix that -i d consumes; feeding it back gives the
    NJ tree of the float-rounded matrix."""
    rng = np.random.default_rng(41)
    n, L = 60, 800
    seqs = _util.synth_alignment(rng, n, L, mean_bl=1e-2, lo=1e-3, hi=5e-2)
    names = [f"T{i+1}" for i in range(n)]
    fa = tmp_path / "a.fa"
    _util.write_fasta(str(fa), names, seqs)
    phy = tmp_path / "d.phy"
    assert run("-i", "m", "-o", "d", "-I", str(fa), "-O", str(phy), "-d", "1").returncode == 0
    lines = phy.read_text().strip().split("\n")
    assert int(lines[0]) == n and [ln.split("\t")[0] for ln in lines[1:]] == names
    D = orc.msa_dist_lower(orc.pack4_many(seqs), L, 1)
    for i in range(n):
        vals = [float(v) for v in lines[1 + i].split("\t")[1:]]
        assert len(vals) == i and np.allclose(vals, D[i, :i], rtol=1e-8, atol=0)
    out = tmp_path / "t.nwk"
    assert run("-i", "d", "-I", str(phy), "-O", str(out)).returncode == 0
    Dr = np.zeros((n, n))
    for i in range(n):
        for j, v in enumerate(lines[1 + i].split("\t")[1:]):
            Dr[i, j] = orc.phylip_value(v)
    ref = orc.nj_run(Dr)
    assert out.read_text() == _util.newick_from_merges(names, ref["merge_x"], ref["merge_y"], ref["bl_x"], ref["bl_y"], ref["last_d"])
