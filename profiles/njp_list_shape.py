"""Where do the units listed by an OLD epoch of the pruned NJ loop lie?  Runs the default plan up to iteration --stop (inside an
epoch), fetches the list of the next scan and the row sums by position (dpr_get_njp_list), and prints
  * units per strip (512 columns) and per row group (16 rows): a few full strips / rows = an outlier row sum in that group;
  * distance of the listed units from the diagonal (in strips): units off the diagonal = loose bounds, not close pairs;
  * the spread of U / (n - 2) inside the row groups and the sub-strips (128 columns) against the step between neighbours.
  python3 profiles/njp_list_shape.py --tips 100000 --stop 18000 [--model gtr+g+i --indel-gaps]"""
import argparse, json, os, subprocess, sys, tempfile
import ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--tips", type=int, default=30000)
ap.add_argument("--sites", type=int, default=10000)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--stop", type=int, nargs="+", default=[5000])
ap.add_argument("--model", default="jc69")
ap.add_argument("--indel-gaps", action="store_true")
args = ap.parse_args()
import numpy as np  # noqa: E402
import dipper_amd  # noqa: E402
from dipper_amd import capi  # noqa: E402

tmp = tempfile.mkdtemp(prefix="nls_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(args.tips), "--sites", str(args.sites), "--seed", str(args.seed), "--packed4", p4]
               + (["--indel-gaps"] if args.indel_gaps else []) + (["--model", args.model] if args.model != "jc69" else []), check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(args.tips, (args.sites + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
L = capi.load_library()
L.dpr_get_njp_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.c_int64]
d = dipper_amd.Dipper(0)
d.set_msa(packed, args.sites)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
done = 0
for stop in sorted(args.stop):
    d.nj_run(max_iters=stop - done)
    done = stop
    cap = 1 << 24
    codes = np.zeros(cap, np.int32); ur = np.zeros(args.tips + 1024, np.float64)
    cnt, P = C.c_int64(0), C.c_int64(0)
    assert L.dpr_get_njp_list(d.h, codes.ctypes.data, cap, C.byref(cnt), C.byref(P), ur.ctypes.data, ur.size) == 0, capi.last_error(L)
    n_list, P = int(cnt.value), int(P.value)
    c = codes[:min(n_list, cap)].view(np.uint32)
    mask, cb, g = c >> 28, (c >> 18) & 1023, c & 0x3FFFF
    live = ~np.isnan(ur[:P])
    rec = {"tips": args.tips, "model": args.model, "indel_gaps": args.indel_gaps, "iteration": stop, "active": args.tips - stop, "positions": P,
           "live_positions": int(live.sum()), "units_listed": n_list, "sub_units_listed": int(np.unpackbits(mask.astype(np.uint8)[:, None], axis=1)[:, 4:].sum())}
    if n_list:
        per_strip = np.bincount(cb, minlength=(P + 511) // 512)
        per_group = np.bincount(g, minlength=(P + 15) // 16)
        off = g // 32 - cb                           # strips between the unit's rows and its columns (0 = on the diagonal)
        rec["strips_total"] = int(per_strip.size); rec["strips_with_units"] = int((per_strip > 0).sum())
        rec["top5_strips_share"] = float(np.sort(per_strip)[-5:].sum() / n_list)
        rec["row_groups_total"] = int(per_group.size); rec["row_groups_with_units"] = int((per_group > 0).sum())
        rec["top50_row_groups_share"] = float(np.sort(per_group)[-50:].sum() / n_list)
        rec["offset_from_diagonal_strips"] = {"0": float((off == 0).mean()), "1-2": float(((off >= 1) & (off <= 2)).mean()),
                                              "3-10": float(((off >= 3) & (off <= 10)).mean()), ">10": float((off > 10).mean())}
    # spread of the row sums inside the groups the bounds are taken over
    u = ur[:P].copy()
    def spread(width):
        m = (P // width) * width
        a = u[:m].reshape(-1, width)
        with np.errstate(all="ignore"):
            hi, lo = np.nanmax(a, axis=1), np.nanmin(a, axis=1)
        ok = ~np.isnan(hi)
        return float(np.median((hi - lo)[ok])), float(np.mean((hi - lo)[ok]))
    lv = np.sort(u[live])
    rec["ur_range"] = [float(lv[0]), float(lv[-1])]
    rec["ur_median_step_between_sorted_neighbours"] = float(np.median(np.diff(lv)))
    rec["ur_spread_in_16_row_group_median_mean"] = spread(16)
    rec["ur_spread_in_128_column_sub_strip_median_mean"] = spread(128)
    # how far is the position order from sorted?  (rank displacement of the live positions)
    order = np.argsort(np.argsort(u[live]))
    disp = np.abs(order - np.arange(order.size))
    disp_r = np.abs(order[::-1] - np.arange(order.size))          # (whichever direction the epoch was sorted in)
    if disp_r.mean() < disp.mean():
        disp = disp_r
    rec["rank_displacement_median_p90_max"] = [float(np.median(disp)), float(np.percentile(disp, 90)), float(disp.max())]
    print(json.dumps(rec), flush=True)
d.close()
