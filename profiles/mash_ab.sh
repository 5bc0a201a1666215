#!/bin/bash
# usage (GPU box): bash profiles/mash_ab.sh [n] [L]  -> pair-distance rate of the three Mash pair kernels at several divergences
N=${1:-20000}; L=${2:-3000}
for bl in 2e-5 1e-4 3e-4 1e-3 1e-2; do
  for mode in "DPR_MASH_KERNEL=noindex" "DPR_MASH_KERNEL=table" "DPR_MASH_KERNEL=index"; do
    echo "== bl=$bl $mode"
    env $mode DPR_LOG=mash python3 profiles/mash_pairs.py $N $L $bl 2>&1 | grep -E "pairs/s|inverted|tokens per" | tail -n 3
  done
done
