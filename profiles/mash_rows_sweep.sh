#!/bin/bash
# usage (GPU box): bash profiles/mash_rows_sweep.sh [tips]  -> Mash distance rows of a placement run (no overlap with the tree kernels) at
# several divergences: distance_wait_ms is the distance kernels' time alone.  Used to choose the index's chunk size (mash_index.hip kIC).
N=${1:-50000}
for bl in 2e-5 1e-3 1e-2 1e-1 1; do
  DPR_PLACE_NO_OVERLAP=1 python3 profiles/place_bench.py $N 3000 r $bl | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('mean branch $bl: distance %.0f ms, tree %.0f ms' % (d['distance_wait_ms'], d['tree_part_ms']))" || exit 1
done
