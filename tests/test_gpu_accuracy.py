"""The reference authors' own acceptance measure (scripts/nrf.sh:26,36-60: normalised Robinson-Foulds distance of the
inferred tree against the simulated TRUE tree), at BASELINE.json's sizes on inputs whose generating tree is known
(tools/gen_synth).  The oracle of this build is a restatement that the reference cannot pin for its CUDA-only parts; this is
the independent guard against a faithful-looking but wrong restatement of NJ, placement, divide-and-conquer and --add:
every case asserts the nRF measured once on an MI355X (the pipeline is deterministic, so the value is reproducible; 0.01 of
slack).  For scale: two unrelated trees give nRF ~ 1; NJ on clean, divergent data ~ 0."""
import json
import os
import sys

import numpy as np
import pytest

from tests import _util

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("DPR_SKIP_FULLSIZE") == "1", reason="DPR_SKIP_FULLSIZE=1"),
              pytest.mark.skipif(not os.path.exists(_util.GEN_SYNTH), reason="tools not built")]

# measured on one MI355X (profiles/r3/nrf_measured.json); None = not pinned yet (the test then only records)
NRF_MEASURED = {
    "nj_30k_bench_clonality": 0.809247591,   # 0.2 substitutions per branch: four of five true splits leave no trace in the data
    "nj_30k_divergent": 0.0701070107,
    "place_100k_mash": 0.213886417,
    "dc_1m": 0.559175678,                    # 400 sites x mean branch 2e-3 = 0.8 substitutions per branch
    "add_50k_onto_500k": 0.629905254,        # 300 sites
}


def _check(name, got):
    rec = os.environ.get("DPR_NRF_RECORD")
    if rec:
        with open(rec, "a") as f:
            f.write(json.dumps({"case": name, **got}) + "\n")
    assert 0.0 <= got["nrf"] <= 1.0
    if NRF_MEASURED[name] is not None:
        assert got["nrf"] <= NRF_MEASURED[name] + 0.01, (name, got)


@pytest.mark.parametrize("name,L,mean", [("nj_30k_bench_clonality", 10000, 2e-5), ("nj_30k_divergent", 2000, 1e-2)])
def test_nj_30k_against_generating_tree(tmp_path, name, L, mean):
    """configs[1]: conventional NJ of 30 000 aligned tips, JC69.  On the bench's near-clonal data (mean branch 2e-5 x
    10 000 sites = 0.2 substitutions per branch) most true branches carry no substitution at all and cannot be recovered
    by ANY method; on divergent data NJ recovers nearly every split."""
    import dipper_amd
    from dipper_amd import capi
    n = 30000
    inp = _util.gen_synth(tmp_path, "a", n, L, 1, mean, mean / 10, mean * 10)
    d = dipper_amd.Dipper(0)
    try:
        d.set_msa(inp["packed4"], L)
        d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        res = d.nj_run()
    finally:
        d.close()
    nwk = _util.newick_from_merges(inp["names"], res["merge_x"], res["merge_y"], res["bl_x"], res["bl_y"], res["last_d"], fmt=repr)
    _check(name, _util.nrf(inp["tree"], nwk, tmp_path))


def test_placement_100k_mash_against_generating_tree(tmp_path):
    """configs[2]: 100 000 unaligned tips (indels), Mash sketches + k-closest placement"""
    import dipper_amd
    from dipper_amd import capi
    n = 100000
    inp = _util.gen_synth(tmp_path, "r", n, 3000, 2, 1e-3, 1e-4, 1e-2, reads=True, shuffle=5)
    d = dipper_amd.Dipper(0)
    try:
        d.set_reads_packed(*inp["reads"])
        d.sketch(15, 1000, fetch=False)
        st = d.place_run(capi.SRC_MASH, n, k=15)
    finally:
        d.close()
    nwk = _util.newick_from_placement(inp["names"], st["head"], st["e"], st["nxt"], st["len"], n, fmt=repr)
    _check("place_100k_mash", _util.nrf(inp["tree"], nwk, tmp_path))


def test_dc_1m_against_generating_tree(tmp_path):
    """configs[3]: divide-and-conquer of 1 000 000 aligned tips (backbone 50 000), input order shuffled like the CLI's"""
    import dipper_amd
    from dipper_amd import capi
    n, L = 1000000, 400
    inp = _util.gen_synth(tmp_path, "d", n, L, 3, 2e-3, 2e-4, 2e-2, shuffle=4)
    d = dipper_amd.Dipper(0)
    try:
        d.set_msa(inp["packed4"], L)
        st = d.dc_run(capi.SRC_MSA, n, n // 20, dist_type=capi.DIST_JC)
    finally:
        d.close()
    nwk = _util.newick_from_placement(inp["names"], st["head"], st["e"], st["nxt"], st["len"], n, fmt=repr)
    _check("dc_1m", _util.nrf(inp["tree"], nwk, tmp_path))


def test_add_50k_onto_500k_against_generating_tree(tmp_path, orc):
    """configs[4]: --add of 50 000 queries onto a 500 000-tip backbone (the backbone: a divide-and-conquer tree of the first
    500 000 tips, written and re-imported like the CLI's -t file); the final tree over all 550 000 tips against the true one"""
    import dipper_amd
    from dipper_amd import capi
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 100000))
    m, nq, L = 500000, 50000, 300
    n = m + nq
    inp = _util.gen_synth(tmp_path, "q", n, L, 5, 2e-3, 2e-4, 2e-2, shuffle=6)
    packed = np.asarray(inp["packed4"])
    names = inp["names"]
    d = dipper_amd.Dipper(0)
    try:
        d.set_msa(packed[:m], L)
        bb = d.dc_run(capi.SRC_MSA, m, m // 20, dist_type=capi.DIST_JC)
        idx_names = ["T%d" % i for i in range(n)]                      # placeholder names = input positions
        nwk = _util.newick_from_placement(idx_names[:m], bb["head"], bb["e"], bb["nxt"], bb["len"], m)
        del bb
        st, leaf_names = _util.backbone_state(orc, nwk, n)          # Tree::Tree ids + adjacency (src/tree.cpp:216-361)
        order = [int(x[1:]) for x in leaf_names] + list(range(m, n))  # backbone tips in import order, then the queries
        d.set_msa(np.ascontiguousarray(packed[order]), L)
        adj = ("head", "e", "nxt", "belong", "len")
        full = d.place_run(capi.SRC_MSA, n, first=m, dist_type=capi.DIST_JC, state={k: st[k].copy() for k in adj})
    finally:
        d.close()
    out_names = [names[i] for i in order]
    nwk = _util.newick_from_placement(out_names, full["head"], full["e"], full["nxt"], full["len"], n, fmt=repr)
    _check("add_50k_onto_500k", _util.nrf(inp["tree"], nwk, tmp_path))
