"""CPU tests of the bench / test infrastructure under tools/: the native synthetic-input generator (gen_synth: seeded
Yule tree, JC69, optional indels, true tree) and the normalised-RF tool (nrf), which implements the reference authors'
accuracy measure (scripts/nrf.sh:26,36-60)."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests import _util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tools", "bin", "gen_synth")
NRF = os.path.join(ROOT, "tools", "bin", "nrf")

pytestmark = pytest.mark.skipif(not (os.path.exists(GEN) and os.path.exists(NRF)), reason="tools not built (python __graft_entry__.py)")


def _fasta(path):
    names, seqs = [], []
    for line in open(path, "rb").read().split(b"\n"):
        if line.startswith(b">"):
            names.append(line[1:].split()[0].decode())
        elif line:
            seqs.append(line)
    return names, seqs


def _nrf(a, b):
    return json.loads(subprocess.run([NRF, a, b], check=True, capture_output=True, text=True).stdout)


def test_gen_synth_outputs_agree_and_do_not_depend_on_threads(tmp_path, orc):
    n, L = 500, 1000
    out = {}
    for tag, threads in (("a", 1), ("b", 5)):
        base = str(tmp_path / tag)
        subprocess.run([GEN, "--tips", str(n), "--sites", str(L), "--seed", "3", "--mean-bl", "1e-2", "--lo", "1e-3", "--hi", "1e-1",
                        "--shuffle", "11", "--threads", str(threads), "--fasta", base + ".fa", "--packed4", base + ".p4",
                        "--tree", base + ".nwk", "--order", base + ".ord"], check=True)
        out[tag] = {k: open(base + k, "rb").read() for k in (".fa", ".p4", ".nwk", ".ord")}
    assert out["a"] == out["b"]
    names, seqs = _fasta(str(tmp_path / "a.fa"))
    order = np.frombuffer(out["a"][".ord"], dtype=np.int32)
    assert sorted(order.tolist()) == list(range(n)) and not np.array_equal(order, np.arange(n))
    assert names == ["T%d" % (k + 1) for k in order]
    assert all(len(s) == L and set(s) <= set(b"ACGT") for s in seqs)
    # --packed4 is fourBitCompressor's encoding of the FASTA rows (the oracle's restatement, src/fourBitCompressor.cpp:5-41)
    packed = np.frombuffer(out["a"][".p4"], dtype=np.uint64).reshape(n, (L + 15) // 16)
    assert np.array_equal(packed, orc.pack4_many(seqs))
    # the true tree is over the same names
    kids, length, name, root = _util.parse_newick(out["a"][".nwk"].decode())
    assert sorted(name[v] for v in name if not kids[v]) == sorted(names)
    assert all(1e-3 <= length[v] <= 1e-1 for v in length)


def test_gen_synth_gap_fraction(tmp_path, orc):
    """--gap-frac: about that fraction of the cells become '-', every other cell and the tree are the gap-free run's, the
    4-bit words carry code 4 there (src/fourBitCompressor.cpp:33-35), and the output does not depend on the thread count"""
    n, L = 300, 3000
    out = {}
    for tag, extra in (("plain", []), ("gaps", ["--gap-frac", "0.03"]), ("gaps5", ["--gap-frac", "0.03", "--threads", "5"]), ("indel", ["--indel-gaps"])):
        base = str(tmp_path / tag)
        subprocess.run([GEN, "--tips", str(n), "--sites", str(L), "--seed", "9", "--mean-bl", "1e-3", "--lo", "1e-4", "--hi", "1e-2",
                        "--fasta", base + ".fa", "--packed4", base + ".p4", "--tree", base + ".nwk"] + extra, check=True)
        out[tag] = {k: open(base + k, "rb").read() for k in (".fa", ".p4", ".nwk")}
    assert out["gaps"] == out["gaps5"]
    assert out["gaps"][".nwk"] == out["plain"][".nwk"]
    _, plain = _fasta(str(tmp_path / "plain.fa"))
    names, gaps = _fasta(str(tmp_path / "gaps.fa"))
    a = np.frombuffer(b"".join(plain), dtype=np.uint8)
    b = np.frombuffer(b"".join(gaps), dtype=np.uint8)
    isgap = b == ord("-")
    assert 0.02 < isgap.mean() < 0.04
    assert np.array_equal(a[~isgap], b[~isgap])
    per_tip = isgap.reshape(n, L).mean(axis=1)
    assert (per_tip > 0).mean() > 0.98 and per_tip.max() < 0.12        # (nearly) every tip has gaps, none is mostly gaps
    packed = np.frombuffer(out["gaps"][".p4"], dtype=np.uint64).reshape(n, (L + 15) // 16)
    assert np.array_equal(packed, orc.pack4_many(gaps))
    # --indel-gaps alone: only the inherited deletions of the authors' indel model -- few cells, shared by clades (a gap cell is
    # far more often shared with another tip than the per-tip runs above), every other cell unchanged
    _, indel = _fasta(str(tmp_path / "indel.fa"))
    c = np.frombuffer(b"".join(indel), dtype=np.uint8).reshape(n, L)
    ig = c == ord("-")
    assert 0 < ig.mean() < 0.01 and out["indel"][".nwk"] == out["plain"][".nwk"]
    assert np.array_equal(a[~ig.ravel()], c.ravel()[~ig.ravel()])
    shared = (ig.sum(axis=0) >= 2)[None, :] & ig
    assert shared.sum() > 0.5 * ig.sum()


def test_gen_synth_reads_with_indels(tmp_path, orc):
    n, L = 200, 2000
    base = str(tmp_path / "r")
    subprocess.run([GEN, "--tips", str(n), "--sites", str(L), "--seed", "5", "--mean-bl", "2e-2", "--lo", "2e-3", "--hi", "2e-1",
                    "--indel", "0.03,0.09", "--fasta", base + ".fa", "--packed2", base, "--tree", base + ".nwk"], check=True)
    names, seqs = _fasta(base + ".fa")
    assert names == ["T%d" % (k + 1) for k in range(n)]
    lens = np.fromfile(base + ".len", dtype=np.uint64)
    off = np.fromfile(base + ".off", dtype=np.uint64)
    flat = np.fromfile(base + ".flat", dtype=np.uint64)
    assert np.array_equal(lens, np.array([len(s) for s in seqs], dtype=np.uint64))
    assert len(set(lens.tolist())) > 1                      # indels change the lengths
    for i in (0, 1, n // 2, n - 1):                         # twoBitCompressor's words (src/twoBitCompressor.cpp:5-41)
        w = orc.pack2(seqs[i])
        assert np.array_equal(flat[int(off[i]):int(off[i]) + len(w)], w)


def _random_newick(rng, names, fmt="%.6g"):
    items = [(nm, None) for nm in names]
    rng.shuffle(items)
    items = [nm for nm, _ in items]
    while len(items) > 1:
        i, j = sorted(rng.choice(len(items), size=2, replace=False).tolist())
        b = items.pop(j)
        a = items.pop(i)
        items.append("(%s:%s,%s:%s)" % (a, fmt % rng.uniform(0, 1), b, fmt % rng.uniform(0, 1)))
    return items[0] + ";\n"


@pytest.mark.parametrize("n", [4, 5, 12, 60, 300])
def test_nrf_tool_equals_bipartition_sets(tmp_path, n):
    """the tool's hashed bipartitions against plain Python sets (tests/_util.splits) on random trees, incl. a tree
    compared with a few-leaf rearrangement of itself and with its own rerooting"""
    rng = np.random.default_rng(n)
    names = ["T%d" % (i + 1) for i in range(n)]
    trees = [_random_newick(rng, names) for _ in range(3)]
    for k, t in enumerate(trees):
        open(tmp_path / ("t%d.nwk" % k), "w").write(t)
    for a in range(3):
        for b in range(3):
            got = _nrf(str(tmp_path / ("t%d.nwk" % a)), str(tmp_path / ("t%d.nwk" % b)))
            sa, sb = _util.splits(trees[a], names), _util.splits(trees[b], names)
            assert got["tips"] == n and got["splits_a"] == len(sa) and got["splits_b"] == len(sb)
            assert got["rf"] == len(sa ^ sb) and got["common"] == len(sa & sb)
            assert got["nrf"] == pytest.approx(len(sa ^ sb) / max(len(sa) + len(sb), 1))
    assert _nrf(str(tmp_path / "t0.nwk"), str(tmp_path / "t0.nwk"))["nrf"] == 0.0


def test_generated_alignment_carries_its_tree(tmp_path, orc):
    """JC69 distances of a long, divergent generated alignment + the oracle's NJ recover the generating tree almost
    completely (nRF small) -- the generator's sequences really evolve down the tree it writes."""
    n, L = 120, 20000
    base = str(tmp_path / "g")
    subprocess.run([GEN, "--tips", str(n), "--sites", str(L), "--seed", "2", "--mean-bl", "2e-2", "--lo", "5e-3", "--hi", "1e-1",
                    "--fasta", base + ".fa", "--tree", base + ".nwk"], check=True)
    names, seqs = _fasta(base + ".fa")
    D = orc.msa_dist_lower(orc.pack4_many(seqs), L, 2)
    res = orc.nj_run(np.tril(D, -1))
    nwk = _util.newick_from_merges(names, res["merge_x"], res["merge_y"], res["bl_x"], res["bl_y"], res["last_d"], fmt=repr)
    open(base + ".nj.nwk", "w").write(nwk)
    got = _nrf(base + ".nwk", base + ".nj.nwk")
    assert got["tips"] == n and got["nrf"] < 0.05


def test_nrf_ignores_the_root(tmp_path):
    """the same unrooted tree written with three different roots (degree-2 root, trifurcation, root next to a tip)"""
    forms = ["((A:1,B:1):1,((C:1,D:1):1,E:1):1);", "((A:1,B:1):2,(C:1,D:1):1,E:1);", "(A:1,(B:1,((C:1,D:1):1,E:1):2):0);"]
    for k, t in enumerate(forms):
        open(tmp_path / ("f%d.nwk" % k), "w").write(t + "\n")
    for a in range(3):
        for b in range(3):
            got = _nrf(str(tmp_path / ("f%d.nwk" % a)), str(tmp_path / ("f%d.nwk" % b)))
            assert got["rf"] == 0 and got["splits_a"] == 2 and got["splits_b"] == 2
    open(tmp_path / "other.nwk", "w").write("((A:1,C:1):1,((B:1,D:1):1,E:1):1);\n")
    got = _nrf(str(tmp_path / "f0.nwk"), str(tmp_path / "other.nwk"))
    assert got["rf"] == 4 and got["nrf"] == 1.0


def test_nrf_rejects_labels_outside_the_tree(tmp_path):
    """malformed / truncated Newick (a label with no enclosing parentheses, text before the first '(') must end in a
    diagnostic and exit code 1, not in an out-of-bounds write (advisor, round 3)"""
    ok = tmp_path / "ok.nwk"
    ok.write_text("((T1:1,T2:1):1,T3:2);\n")
    for text in ("T1:0.1;\n", "x((T1:1,T2:1):1,T3:2);\n", "((T1:1,T2:1):1,T3:2);(T1,T2);\n"):
        bad = tmp_path / "bad.nwk"
        bad.write_text(text)
        r = subprocess.run([NRF, str(bad), str(ok)], capture_output=True, text=True)
        assert r.returncode == 1 and r.stderr.startswith("nrf: "), (text, r.returncode, r.stderr)


def test_gen_synth_gtr_g_i_model(tmp_path):
    """--model gtr+g+i (the substitution model of scripts/alisim.sh:14 with the explicit parameters of its line 21): base
    composition F{0.3,0.2,0.2,0.3}, a fifth of the sites invariant plus the slow gamma class, C<->T and G<->T exchanges far
    ahead of the others, same bytes for any thread count, and the JC69 default untouched."""
    n, L = 64, 40000
    base = str(tmp_path / "g")
    args = [GEN, "--tips", str(n), "--sites", str(L), "--seed", "5", "--mean-bl", "0.05", "--lo", "0.01", "--hi", "0.2"]
    subprocess.run(args + ["--model", "gtr+g+i", "--threads", "1", "--fasta", base + "1.fa"], check=True)
    subprocess.run(args + ["--model", "GTR+G+I", "--threads", "4", "--fasta", base + "4.fa"], check=True)
    assert open(base + "1.fa", "rb").read() == open(base + "4.fa", "rb").read()
    _, seqs = _fasta(base + "1.fa")
    A = np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(n, L)
    freq = {c: float((A == ord(c)).mean()) for c in "ACGT"}
    for c, f in zip("ACGT", (0.3, 0.2, 0.2, 0.3)):
        assert abs(freq[c] - f) < 0.02, freq
    constant = float((A == A[0]).all(axis=0).mean())
    assert 0.2 <= constant < 0.5, constant            # 20 % invariant + most of the slowest gamma class
    # exchanges between the first tip and every other one, by unordered pair of bases (short branches: hardly any site hit twice)
    subprocess.run([GEN, "--tips", str(n), "--sites", str(L), "--seed", "5", "--mean-bl", "0.001", "--lo", "0.0002", "--hi", "0.004",
                    "--model", "gtr+g+i", "--fasta", base + "s.fa"], check=True)
    _, ss = _fasta(base + "s.fa")
    B = np.frombuffer(b"".join(ss), dtype=np.uint8).reshape(n, L)
    pairs = {}
    for r in range(1, n):
        d = B[0] != B[r]
        for x, y in zip(B[0][d], B[r][d]):
            k = "".join(sorted(chr(x) + chr(y)))
            pairs[k] = pairs.get(k, 0) + 1
    tot = sum(pairs.values())
    assert (pairs.get("CT", 0) + pairs.get("GT", 0)) / tot > 0.75, pairs
    assert pairs.get("CG", 0) / tot < 0.02 and pairs.get("AC", 0) / tot < 0.03, pairs
    # the default model is still JC69: uniform composition, every exchange alike
    subprocess.run(args + ["--fasta", base + "j.fa"], check=True)
    subprocess.run(args + ["--model", "jc69", "--fasta", base + "j2.fa"], check=True)
    assert open(base + "j.fa", "rb").read() == open(base + "j2.fa", "rb").read()
    _, sj = _fasta(base + "j.fa")
    J = np.frombuffer(b"".join(sj), dtype=np.uint8).reshape(n, L)
    for c in "ACGT":
        assert abs(float((J == ord(c)).mean()) - 0.25) < 0.02
    r = subprocess.run([GEN, "--tips", "10", "--sites", "100", "--model", "hky", "--fasta", base + "x.fa"], capture_output=True, text=True)
    assert r.returncode != 0 and "--model" in r.stderr


def test_oracle_under_address_and_ub_sanitizers():
    """the CPU oracle built with -fsanitize=address,undefined (`make -C oracle asan`) runs its own test file in a
    child process with libasan preloaded: no report.  CPU only (GPU sanitizers are not available on this pool)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    odir = os.path.join(root, "oracle")
    r = subprocess.run(["make", "-C", odir, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(libasan), libasan
    env = dict(os.environ, DPR_ORACLE_LIB=os.path.join(odir, "liboracle_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:exitcode=66", UBSAN_OPTIONS="halt_on_error=1:exitcode=66", OMP_NUM_THREADS="4")
    q = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_oracle.py")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    tail = (q.stdout + q.stderr)[-3000:]
    assert q.returncode == 0 and "passed" in q.stdout and "Sanitizer" not in tail and "runtime error" not in tail, tail
