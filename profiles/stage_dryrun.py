"""Dry run of bench.py's input staging for an 8-rank job on a box WITHOUT GPUs (verdict r2 item 1: "a dry run of the 8-rank
input path on the CPU box (8 processes, no GPU calls) showing the wall time of input staging"): 8 processes joined by
gloo build bench.Stage, rank 0 generates every input bench.py uses once (tools/bin/gen_synth), all ranks map them and
checksum what they see.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 profiles/stage_dryrun.py"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.distributed as dist

import bench

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
dist.init_process_group("gloo")
t0 = time.perf_counter()
stage = bench.Stage(rank, world, dist)
rec = {"world": world, "host_cores": bench.host_cores(), "threads_per_rank": max(1, bench.host_cores() // world), "inputs": {}}
try:
    for tag, kw in (("main", dict(tips=30000, sites=10000, seed=1, mean=2e-5, lo=2e-6, hi=2e-4, fasta=True)),
                    ("nj100k", dict(tips=100000, sites=10000, seed=8, mean=2e-5, lo=2e-6, hi=2e-4)),
                    ("dc1m", dict(tips=1000000, sites=400, seed=10, mean=2e-3, lo=2e-4, hi=2e-2, shuffle=7))):
        ts = time.perf_counter()
        p = stage.gen(tag, **kw)
        gen_s = time.perf_counter() - ts
        ts = time.perf_counter()
        a = bench.Stage.packed4(p)
        h = hashlib.sha256(np.ascontiguousarray(a[:: max(1, a.shape[0] // 1000)]).tobytes()).hexdigest()[:12]     # a sample of the rows
        hs = [None] * world
        dist.all_gather_object(hs, h)
        rec["inputs"][tag] = {"tips": kw["tips"], "sites": kw["sites"], "generate_and_wait_s": round(gen_s, 2), "map_and_sample_s": round(time.perf_counter() - ts, 3),
                              "bytes_packed4": os.path.getsize(p["packed4"]), "bytes_fasta": os.path.getsize(p["fasta"]) if p["fasta"] else 0,
                              "all_ranks_see_the_same_rows": len(set(hs)) == 1}
    rec["staging_wall_s"] = round(time.perf_counter() - t0, 2)
    rec["directory"] = os.path.dirname(stage.dir)
finally:
    stage.sync()
    stage.cleanup()
if rank == 0:
    print(json.dumps(rec))
dist.destroy_process_group()
