"""The distance block of the divide-and-conquer assignment / --add at the authors' sequence length, alone:
    python3 profiles/msa_block_bench.py [--queries 5120] [--backbone 50000] [--sites 10000] [--reps 5] [--gap]
prints pairs/s, site-pairs/s and lane-ops/s (7 integer operations per 32 sites and pair) of msa_dist_kernel<JC> through
dpr_msa_dist_block.  The profiling target for `rocprofv3 --pmc` passes on that kernel (profiles/prof.sh pmc ...)."""
import argparse, json, os, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=5120)
ap.add_argument("--backbone", type=int, default=50000)
ap.add_argument("--sites", type=int, default=10000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--gap", action="store_true", help="inherited deletion gaps (gen_synth --indel-gaps), as the bench's inputs")
ap.add_argument("--gap-frac", type=float, default=0.0)
args = ap.parse_args()
import numpy as np
import dipper_amd
from dipper_amd import capi
n, L = args.backbone + args.queries, args.sites
tmp = tempfile.mkdtemp(prefix="msab_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "4", "--mean-bl", "2e-4", "--lo", "2e-5",
                "--hi", "2e-3", "--model", "gtr+g+i", "--packed4", p4] + (["--indel-gaps"] if args.gap or args.gap_frac > 0 else [])
               + (["--gap-frac", repr(args.gap_frac)] if args.gap_frac > 0 else []), check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
_, ms = d.msa_dist_block(args.backbone, args.queries, args.backbone, dist_type=2, transposed=True, fetch=False, reps=args.reps)
pairs = args.queries * args.backbone
words = (L + 31) // 32
print(json.dumps({"queries": args.queries, "backbone": args.backbone, "sites": L, "ms_per_block": ms, "pairs_per_s": pairs / (ms * 1e-3),
                  "site_pairs_per_s": pairs * L / (ms * 1e-3), "lane_ops_per_s_at_7_per_word": pairs * words * 7 / (ms * 1e-3)}))
d.close()
