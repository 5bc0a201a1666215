#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only (needs
/root/reference and oracle/_ref built from the reference's own src/tree.cpp):

    make -C oracle && python tests/golden/make_golden.py

Fixtures
  t2.backbone.nwk            the reference's only data file (dataset/t2.backbone.nwk), verbatim
  t2_ref_tree.npz            flattening of Tree::Tree(newick, 10000) as built by the REFERENCE's
                             own tree.cpp: per node (pre-order) idx, parent idx, branch length,
                             leaf flag, name  -> pins the Newick import order of a-11
  survey_known_answers.json  values captured from the reference's host objects during the survey
                              + public MurmurHash3 vectors
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def ref_tree():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_tree.so"))
    nwk = open(os.path.join(HERE, "t2.backbone.nwk")).readline().strip()
    cap = 4096
    idx = np.zeros(cap, np.int32); par = np.zeros(cap, np.int32); bl = np.zeros(cap, np.float64)
    leaf = np.zeros(cap, np.int32); names = C.create_string_buffer(64 * cap)
    lib.ref_tree_flatten.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    n = lib.ref_tree_flatten(nwk.encode(), 10000, cap, idx.ctypes.data, par.ctypes.data, bl.ctypes.data,
                             leaf.ctypes.data, names)
    assert n > 0
    nm = [names.raw[64 * i:64 * i + 64].split(b"\0")[0].decode() for i in range(n)]
    np.savez_compressed(os.path.join(HERE, "t2_ref_tree.npz"), idx=idx[:n], parent=par[:n], bl=bl[:n],
                        is_leaf=leaf[:n], name=np.array(nm))
    print("t2_ref_tree.npz:", n, "nodes,", int(leaf[:n].sum()), "leaves; root idx", idx[0], "first leaf", nm[[i for i in range(n) if leaf[i]][0]])


def known_answers():
    ka = {
        "provenance": "SURVEY.md Appendix A: captured from the reference's own host objects "
                      "(src/divide_and_conquer/mash.cpp, src/tree.cpp) during the survey, plus public MurmurHash3 vectors",
        "murmur3_x64_128": [
            {"data": "hello", "seed": 0, "h1": "0xcbd8a7b341bd9b02", "h2": "0x5b1e906a48ae1d19"},
            {"data": "AAAAAAAAAAAAAAA", "seed": 42, "h1": "0xf3e87bc255d2a127"},
            {"data": "ACGTACGTACGTACG", "seed": 42, "h1": "0x456b3e2e10c981c6"},
            {"data": "AC", "seed": 42, "h1": "0xd51c0479c4f743fa"},
        ],
        "mash_dist_k15_S1000": [
            {"row": "2*i+10", "col": "2*i+10", "d": 0.0},
            {"row": "2*i+10", "col": "2*i+10 if i<500 else 2*i+11", "d": 0.027031007207210963},
            {"row": "7+i", "col": "7", "d": 0.41437383991701832},
            {"row": "7", "col": "7+i", "d": 0.0},
        ],
        "t2_backbone_tree": {"nodes": 1999, "leaves": 1000, "root_idx": 10000, "root_children": 2,
                             "first_leaf": "T9326", "first_leaf_idx": 0, "first_leaf_parent_idx": 10006},
    }
    json.dump(ka, open(os.path.join(HERE, "survey_known_answers.json"), "w"), indent=1)


if __name__ == "__main__":
    ref_tree()
    known_answers()
