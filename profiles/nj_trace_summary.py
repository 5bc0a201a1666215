"""Summarise a rocprofv3 --kernel-trace CSV of an NJ run: per kernel the distribution of the dispatch durations and of the
gaps to the previous dispatch, overall and per block of 2 000 dispatches (= how an iteration's cost moves along the run).
usage: python3 profiles/nj_trace_summary.py <dir or kernel_trace.csv> [out.json]"""
import csv, glob, json, os, sys
import numpy as np

src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = {}
for s, e, k in rows:
    short = k.split("(")[0].replace("void dpr::", "").replace("dpr::", "")
    names.setdefault(short, []).append((s, e))
out = {}
prev_end = None
gaps = {}
for s, e, k in rows:
    short = k.split("(")[0].replace("void dpr::", "").replace("dpr::", "")
    if prev_end is not None:
        gaps.setdefault(short, []).append(s - prev_end)
    prev_end = e
for k, v in sorted(names.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    d = np.array([e - s for s, e in v], dtype=np.float64) / 1e3
    g = np.array(gaps.get(k, [0]), dtype=np.float64) / 1e3
    rec = {"calls": len(v), "total_ms": float(d.sum() / 1e3), "us": {p: float(np.percentile(d, q)) for p, q in (("min", 0), ("p10", 10), ("median", 50), ("p90", 90), ("p99", 99), ("max", 100))},
           "mean_us": float(d.mean()), "gap_before_us": {"median": float(np.median(g)), "mean": float(g[g < 1000].mean()) if (g < 1000).any() else None}}
    if len(v) > 4000:
        rec["mean_us_per_2000_calls"] = [round(float(d[i:i + 2000].mean()), 2) for i in range(0, len(d), 2000)]
        rec["p90_us_per_2000_calls"] = [round(float(np.percentile(d[i:i + 2000], 90)), 2) for i in range(0, len(d), 2000)]
    out[k] = rec
    print(f"{k[:44]:44s} calls {len(v):6d}  total {rec['total_ms']:9.2f} ms  mean {rec['mean_us']:8.2f} us  median {rec['us']['median']:8.2f}  p90 {rec['us']['p90']:8.2f}  p99 {rec['us']['p99']:8.2f}  gap-before median {rec['gap_before_us']['median']:6.2f}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
