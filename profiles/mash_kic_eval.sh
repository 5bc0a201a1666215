#!/bin/bash
# usage (GPU box): bash profiles/mash_kic_eval.sh  -> the three Mash-bound timings a chunk size (mash_index.hip kIC) is judged by
python3 profiles/add_bench.py 500000 50000 1000 r | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('add 50k onto 500k: distance %.0f ms, add %.2f s, backbone tree %.2f s' % (d['distance_wait_ms'], d['add_s'], d['backbone_tree_s']))" || exit 1
for bl in 2e-5 1e-3; do
python3 profiles/place_bench.py 100000 3000 r $bl | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('place 100k, mean branch $bl: %.3f s (distance busy %.0f + wait %.0f ms)' % (d['placement_s'], d['distance_busy_ms'], d['distance_wait_ms']))" || exit 1
done
