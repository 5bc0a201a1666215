# A/B of msa.hip build variants (_exp/lib_*.so, built by hand) on ONE box, over gap densities
cd $GRAFT_REPO_ROOT
for lib in _exp/lib_*.so; do
  for a in "" "--gap" "--gap-frac 0.0005" "--gap-frac 0.03"; do
    echo "== $lib $a $(DPR_LIB=$GRAFT_REPO_ROOT/$lib python3 profiles/msa_block_bench.py $a --reps 5 | grep -o 'ms_per_block": [0-9.]*')"
  done
done
