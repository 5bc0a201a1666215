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
