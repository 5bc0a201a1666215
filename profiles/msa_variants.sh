# A/B of msa.hip builds (_exp/lib_*.so, built by hand) on ONE box, over alignment lengths and gap densities
cd $GRAFT_REPO_ROOT
for lib in _exp/lib_*.so; do
  for a in "--gap --sites 400" "--gap --sites 1000" "--gap --sites 2000" "--gap --sites 10000" "--sites 10000"; do
    echo "== $lib $a $(DPR_LIB=$GRAFT_REPO_ROOT/$lib python3 profiles/msa_block_bench.py $a --reps 10 | grep -o 'ms_per_block": [0-9.]*')"
  done
done
