"""One-off cross-check at scale: k-closest placement of n unaligned tips with the inverted-index kernel and with the bucket-table
kernel (round 1) must give the same trace (winning edge, split position, pendant length of every tip) -- i.e. every distance the
placement read was the same double.  python profiles/mash_kernels_agree.py [n] [mean branch]"""
import os, sys, time, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
mbl = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
seqs = _util.synth_reads(np.random.default_rng(11), n, 3000, mean_bl=mbl, lo=mbl / 10, hi=mbl * 10)
seqs = [seqs[i] for i in np.random.default_rng(12).permutation(n)]
res = {}
for name, env in (("index", {"DPR_MASH_KERNEL": "index"}), ("table", {"DPR_MASH_KERNEL": "table"})):
    for k in ("DPR_MASH_KERNEL",):
        os.environ.pop(k, None)
    os.environ.update(env)
    d = dipper_amd.Dipper(0)
    d.set_reads(seqs)
    d.sketch(15, 1000, fetch=False)
    t0 = time.perf_counter()
    st = d.place_run(capi.SRC_MASH, n, k=15)
    dt = time.perf_counter() - t0
    h = hashlib.sha256()
    live = 4 * n - 4                                   # slots in use; the arrays are sized 8 n and not initialised beyond
    for key in ("head", "e", "nxt", "belong", "len", "trace"):
        a = st[key] if key in ("head", "trace") else st[key][:live]
        h.update(np.ascontiguousarray(a).tobytes())
    res[name] = (h.hexdigest()[:16], st["trace"].copy(), {key: np.ascontiguousarray(st[key]).copy() for key in ("head", "e", "nxt", "belong", "len")})
    print(f"{name}: placement of {n} tips (mean branch {mbl}) in {dt:.2f} s, digest {res[name][0]}", flush=True)
    d.close()
same = res["index"][0] == res["table"][0]
print("identical adjacency + trace:", same)
if not same:
    bad = np.nonzero((res["index"][1] != res["table"][1]).any(axis=1))[0]
    print("first differing tips:", bad[:10])
    for key in ("head", "e", "nxt", "belong", "len"):
        a, b = res["index"][2][key], res["table"][2][key]
        if key != "head": a, b = a[:4 * n - 4], b[:4 * n - 4]
        w = np.nonzero(a != b)[0]
        print(key, len(w), w[:6], a[w[:3]], b[w[:3]], "of", len(a))
sys.exit(0 if same else 1)
