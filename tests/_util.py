"""Shared test helpers: seeded synthetic inputs (no alisim in the image), Newick
assembly mirroring the reference's print routine, patristic distances."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def yule_tree(rng, n):
    """Random Yule-Harding topology: returns (parent, children) over 2n-1 nodes, root 0,
    and the list of leaf node ids in creation order."""
    parent = [-1]
    children = [[]]
    leaves = [0]
    while len(leaves) < n:
        k = int(rng.integers(len(leaves)))
        node = leaves[k]
        a, b = len(parent), len(parent) + 1
        parent += [node, node]
        children += [[], []]
        children[node] = [a, b]
        leaves[k] = a
        leaves.append(b)
    return parent, children, leaves


def synth_alignment(rng, n, L, mean_bl=2e-4, lo=2e-5, hi=2e-3, invalid_frac=0.0):
    """JC69 evolution down a Yule tree (exponential branch lengths clipped to [lo,hi], cf.
    scripts/alisim.sh:14 `-rlen`), no indels.  Returns a list of n byte strings over ACGT
    (plus '-'/'N' when invalid_frac > 0)."""
    parent, children, leaves = yule_tree(rng, n)
    nn = len(parent)
    seq = [None] * nn
    seq[0] = rng.integers(0, 4, size=L, dtype=np.uint8)
    pending = [0]
    out = {}
    while pending:
        node = pending.pop()
        s = seq[node]
        if not children[node]:
            out[node] = s
            seq[node] = None
            continue
        for c in children[node]:
            bl = float(np.clip(rng.exponential(mean_bl), lo, hi))
            t = s.copy()
            k = rng.poisson(L * bl)
            if k:
                pos = rng.integers(0, L, size=k)
                t[pos] = (t[pos] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
            seq[c] = t
            pending.append(c)
        seq[node] = None
    res = []
    for leaf in leaves:
        s = BASES[out[leaf]]
        if invalid_frac > 0:
            s = s.copy()
            m = rng.random(L) < invalid_frac
            s[m] = rng.choice(np.frombuffer(b"-Nacgt", dtype=np.uint8), size=int(m.sum()))
        res.append(s.tobytes())
    return res


def synth_reads(rng, n, L, mean_bl=2e-4, lo=2e-5, hi=2e-3, ins_rate=0.03, del_rate=0.09, mean_indel=2.0):
    """Unaligned counterpart of synth_alignment for the Mash inputs: the same JC69 substitutions down a Yule tree plus
    insertions and deletions at `ins_rate` / `del_rate` events per substitution (scripts/alisim.sh:14 of the reference
    passes `--indel 0.03,0.09` to iqtree2's alisim), geometric lengths with mean `mean_indel`.  Returns n byte strings
    of different lengths."""
    parent, children, leaves = yule_tree(rng, n)
    nn = len(parent)
    seq = [None] * nn
    seq[0] = rng.integers(0, 4, size=L, dtype=np.uint8)
    pending = [0]
    out = {}
    pgeo = 1.0 / max(mean_indel, 1.0)
    while pending:
        node = pending.pop()
        s = seq[node]
        if not children[node]:
            out[node] = s
            seq[node] = None
            continue
        for c in children[node]:
            bl = float(np.clip(rng.exponential(mean_bl), lo, hi))
            t = s.copy()
            k = rng.poisson(len(t) * bl)
            if k:
                pos = rng.integers(0, len(t), size=k)
                t[pos] = (t[pos] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
            for _ in range(rng.poisson(len(t) * bl * del_rate)):
                m = int(rng.geometric(pgeo))
                if len(t) > m + 32:
                    p0 = int(rng.integers(0, len(t) - m))
                    t = np.delete(t, np.s_[p0:p0 + m])
            for _ in range(rng.poisson(len(t) * bl * ins_rate)):
                m = int(rng.geometric(pgeo))
                p0 = int(rng.integers(0, len(t) + 1))
                t = np.insert(t, p0, rng.integers(0, 4, size=m, dtype=np.uint8))
            seq[c] = t
            pending.append(c)
        seq[node] = None
    return [BASES[out[leaf]].tobytes() for leaf in leaves]


def random_additive_matrix(rng, n, zero_frac=0.0):
    """Patristic distance matrix of a random binary tree with n tips (exact tree metric up to fp
    rounding).  zero_frac of the branches get length 0 (ties, as in near-clonal data)."""
    parent, children, leaves = yule_tree(rng, n)
    nn = len(parent)
    bl = rng.uniform(0.01, 1.0, size=nn)
    if zero_frac > 0:
        bl[rng.random(nn) < zero_frac] = 0.0
    bl[0] = 0.0
    depth_path = []
    for leaf in leaves:
        path = {}
        d = 0.0
        v = leaf
        while v != -1:
            path[v] = d
            d += bl[v]
            v = parent[v]
        depth_path.append(path)
    D = np.zeros((n, n))
    for i in range(n):
        pi = depth_path[i]
        for j in range(i):
            pj = depth_path[j]
            v = leaves[j]
            while v not in pi:
                v = parent[v]
            D[i, j] = D[j, i] = pi[v] + pj[v]
    return D


def fmt(v):
    """std::ostream << double with default precision (6 significant digits, %g style)."""
    return "%g" % v


def newick_from_merges(names, mx, my, bx, by, last_d, fmt=fmt):
    """Mirror of the bookkeeping + print of src/neighborJoining.cu:233-270."""
    N = len(names)
    real = list(range(N))
    kids = {}
    ID = N
    for it in range(N - 2):
        x, y = int(mx[it]), int(my[it])
        kids[ID] = [(real[x], float(bx[it])), (real[y], float(by[it]))]
        real[x] = ID
        ID += 1
        real[y] = real[N - it - 1]
    kids[2 * N - 2] = [(real[0], last_d * 0.5), (real[1], last_d * 0.5)]
    out = []
    stack = [("node", 2 * N - 2)]
    while stack:
        kind, v = stack.pop()
        if kind == "text":
            out.append(v)
            continue
        if v in kids:
            out.append("(")
            items = kids[v]
            seq = []
            for i, (c, l) in enumerate(items):
                seq.append(("node", c))
                seq.append(("text", ":" + fmt(l) + (")" if i + 1 == len(items) else ",")))
            stack.extend(reversed(seq))
        else:
            out.append(names[v])
    return "".join(out) + ";\n"


def parse_newick(s):
    """Minimal Newick parser -> (children dict, length dict, name dict, root id)."""
    s = s.strip().rstrip(";")
    kids, length, name = {}, {}, {}
    nid = [0]

    def new():
        nid[0] += 1
        return nid[0] - 1

    # iterative parser
    cur = None
    stack = []
    i = 0
    n = len(s)
    root = None
    while i < n:
        c = s[i]
        if c == "(":
            v = new()
            kids[v] = []
            if stack:
                kids[stack[-1]].append(v)
            else:
                root = v
            stack.append(v)
            cur = None
            i += 1
        elif c == ",":
            cur = None
            i += 1
        elif c == ")":
            cur = stack.pop()
            i += 1
        elif c == ":":
            j = i + 1
            while j < n and s[j] not in ",()":
                j += 1
            length[cur] = float(s[i + 1:j])
            i = j
        else:
            j = i
            while j < n and s[j] not in ":,()":
                j += 1
            label = s[i:j]
            if cur is None:
                v = new()
                kids[v] = []
                name[v] = label
                if stack:
                    kids[stack[-1]].append(v)
                else:
                    root = v
                cur = v
            else:
                name[cur] = label
            i = j
    return kids, length, name, root


def patristic(newick, names):
    kids, length, name, root = parse_newick(newick)
    parent = {}
    for p, cs in kids.items():
        for c in cs:
            parent[c] = p
    leaf_of = {name[v]: v for v in name if not kids[v]}
    paths = []
    for nm in names:
        v = leaf_of[nm]
        d = 0.0
        path = {}
        while True:
            path[v] = d
            if v not in parent:
                break
            d += length.get(v, 0.0)
            v = parent[v]
        paths.append(path)
    n = len(names)
    D = np.zeros((n, n))
    for i in range(n):
        for j in range(i):
            v = leaf_of[names[j]]
            while v not in paths[i]:
                v = parent[v]
            D[i, j] = D[j, i] = paths[i][v] + paths[j][v]
    return D


def splits(newick, names):
    """Set of non-trivial bipartitions (as frozensets of the side not containing names[0])."""
    kids, length, name, root = parse_newick(newick)
    idx = {nm: i for i, nm in enumerate(names)}
    res = set()
    below = {}
    order = []
    st = [root]
    while st:
        v = st.pop()
        order.append(v)
        st.extend(kids[v])
    for v in reversed(order):
        if not kids[v]:
            below[v] = frozenset([idx[name[v]]])
        else:
            below[v] = frozenset().union(*[below[c] for c in kids[v]])
    full = frozenset(range(len(names)))
    for v, b in below.items():
        if 1 < len(b) < len(names) - 1:
            res.add(b if 0 not in b else full - b)
    return res


def newick_from_placement(names, head, e, nxt, length, N, fmt=fmt):
    """Mirror of KPlacementDeviceArrays::printTree (src/placement_close_k.cu:568-643): root = node
    N, children in adjacency-list order, the edge back to the parent skipped."""
    out = []
    stack = [("node", N, -1)]
    while stack:
        item = stack.pop()
        if item[0] == "text":
            out.append(item[1])
            continue
        _, node, frm = item
        if nxt[head[node]] != -1:
            out.append("(")
            pos = []
            i = head[node]
            while i != -1:
                if e[i] != frm:
                    pos.append(i)
                i = nxt[i]
            seq = []
            for k, p in enumerate(pos):
                seq.append(("node", int(e[p]), node))
                seq.append(("text", ":" + fmt(float(length[p])) + (")" if k + 1 == len(pos) else ",")))
            stack.extend(reversed(seq))
        else:
            out.append(names[node])
    return "".join(out) + ";\n"


def ref_findmin_emulation(D, U, n):
    """Literal (slow) emulation of findMinDist<<<256,256>>> + thrust::min_element
    (src/neighborJoining.cu:117-148,214): returns the winning tuple (x, y, q)."""
    gs = bs = 256
    best = None
    r = float(n - 2)
    for bx in range(gs):
        sz = n // gs
        st = sz * bx
        if n % gs > bx:
            sz += 1
        st += min(bx, n % gs)
        ed = st + sz
        if sz == 0:
            continue
        for tx in range(bs):
            minD, x, y = 10000.0, 0, 0
            for j in range(tx, n, bs):
                colU = U[j] / r
                for i in range(st, ed):
                    temp = D[i, j] - U[i] / r - colU
                    if i != j and temp < minD:
                        minD, x, y = temp, i, j
            if best is None or minD < best[2]:
                best = (x, y, minD)
    return best


def write_phylip_lower(path, names, D, sep="\t", digits=9):
    with open(path, "w") as f:
        f.write(f"{len(names)}\n")
        for i, nm in enumerate(names):
            f.write(nm)
            for j in range(i):
                f.write(sep + ("%.*g" % (digits, D[i, j])))
            f.write("\n")


def write_fasta(path, names, seqs, width=0):
    with open(path, "wb") as f:
        for nm, s in zip(names, seqs):
            f.write(b">" + nm.encode() + b" some comment\n")
            if width:
                for k in range(0, len(s), width):
                    f.write(s[k:k + width] + b"\n")
            else:
                f.write(s + b"\n")


def backbone_state(orc, newick, N):
    """Mirror of Tree::Tree id assignment (src/tree.cpp:216-361) + initializeDeviceArrays
    (src/placement_close_k.cu:126-264): returns (state arrays, backbone leaf names by idx)."""
    kids, length, name, root = parse_newick(newick)
    order = sorted(kids.keys())          # creation order: '(' and leaf labels as they appear
    idx, nl, ni = {}, 0, 0
    for v in order:
        if kids[v]:
            idx[v] = N + ni
            ni += 1
        else:
            idx[v] = nl
            nl += 1
    parent = {c: p for p, cs in kids.items() for c in cs}
    bl = {v: float(np.float32(length.get(v, 0.0))) for v in order}
    bl[root] = 0.0
    st = orc.place_alloc(N)
    ec = 0
    stack = [(root, 0)]
    while stack:
        v, k = stack.pop()
        if k < len(kids[v]):
            stack.append((v, k + 1))
            stack.append((kids[v][k], 0))
            continue
        if v in parent:
            x, y = idx[v], idx[parent[v]]
            for (src, dst) in ((x, y), (y, x)):
                st["e"][ec] = dst; st["len"][ec] = bl[v]; st["belong"][ec] = src
                st["nxt"][ec] = st["head"][src]; st["head"][src] = ec
                ec += 1
    leaf_names = [None] * nl
    for v in order:
        if not kids[v]:
            leaf_names[idx[v]] = name[v]
    return st, leaf_names


# ---- native generator / normalised RF (tools/) -------------------------------------------------------------------------
import json as _json
import os as _os
import subprocess as _subprocess

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
GEN_SYNTH = _os.path.join(_ROOT, "tools", "bin", "gen_synth")
NRF_TOOL = _os.path.join(_ROOT, "tools", "bin", "nrf")


def gen_synth(outdir, tag, tips, sites, seed, mean_bl, lo, hi, reads=False, shuffle=None, fasta=False, extra=()):
    """Seeded synthetic input with its generating tree (tools/gen_synth.cpp).  Returns a dict: tree (path), names (tip
    names in OUTPUT order), packed4 ([tips][ceil(sites/16)] uint64 memmap) or reads (flat, off, len), fasta (path)."""
    base = _os.path.join(str(outdir), tag)
    cmd = [GEN_SYNTH, "--tips", str(tips), "--sites", str(sites), "--seed", str(seed), "--mean-bl", repr(mean_bl), "--lo", repr(lo),
           "--hi", repr(hi), "--tree", base + ".nwk", "--order", base + ".ord"]
    if reads:
        cmd += ["--indel", "0.03,0.09", "--packed2", base]
    else:
        cmd += ["--packed4", base + ".p4"]
    if shuffle is not None:
        cmd += ["--shuffle", str(shuffle)]
    if fasta:
        cmd += ["--fasta", base + ".fa"]
    cmd += list(extra)            # e.g. ("--model", "gtr+g+i", "--indel-gaps")
    _subprocess.run(cmd, check=True)
    order = np.fromfile(base + ".ord", dtype=np.int32)
    out = {"tree": base + ".nwk", "names": ["T%d" % (k + 1) for k in order], "fasta": base + ".fa" if fasta else None}
    if reads:
        out["reads"] = tuple(np.fromfile(base + ext, dtype=np.uint64) for ext in (".flat", ".off", ".len"))
    else:
        out["packed4"] = np.memmap(base + ".p4", dtype=np.uint64, mode="r", shape=(tips, (sites + 15) // 16))
    return out


def nrf(true_tree_path, newick_text, workdir, tag="t"):
    """normalised Robinson-Foulds distance of a Newick text against a tree file (tools/nrf.cpp; the reference authors'
    accuracy measure, scripts/nrf.sh:26,36-60)"""
    path = _os.path.join(str(workdir), "nrf_%s.nwk" % tag)
    with open(path, "w") as f:
        f.write(newick_text)
    r = _subprocess.run([NRF_TOOL, true_tree_path, path], check=True, capture_output=True, text=True)
    _os.unlink(path)
    return _json.loads(r.stdout)


def reads_prefix(reads, m):
    """the first m reads of a packed (flat, off, len) triple"""
    flat, off, ln = reads
    end = int(off[m]) if m < len(off) else len(flat)
    return flat[:end], off[:m], ln[:m]


def reads_reorder(reads, order):
    """packed reads in another order: read k of the result is read order[k] of the input"""
    flat, off, ln = reads
    order = np.asarray(order, dtype=np.int64)
    nw = (ln.astype(np.int64) + 31) // 32
    new_nw = nw[order]
    new_off = np.zeros(len(order), dtype=np.uint64)
    new_off[1:] = np.cumsum(new_nw)[:-1].astype(np.uint64)
    # gather the words: index of every output word in the input array
    starts = off.astype(np.int64)[order]
    tot = int(new_nw.sum())
    idx = np.repeat(starts - np.concatenate(([0], np.cumsum(new_nw)[:-1])), new_nw) + np.arange(tot)
    return np.ascontiguousarray(flat[idx]), new_off, np.ascontiguousarray(ln[order])
