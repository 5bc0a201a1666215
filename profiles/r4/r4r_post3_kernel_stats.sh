# round 4, call r: per-kernel averages of the post3 + cells variant (commit 9d05741, staged under _exp/post3) next to the tree's
O=gpurun_out/r4/r; mkdir -p $O
R=$GRAFT_REPO_ROOT
bash profiles/prof.sh trace fused_now python3 profiles/nj_target.py --reps 1 2>&1 | grep -E "njp_|nj_ms"
export GRAFT_REPO_ROOT=$R/_exp/post3
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
bash $R/_exp/post3/profiles/prof.sh trace post3 python3 profiles/nj_target.py --reps 1 2>&1 | grep -E "njp_|nj_ms"
cp -r $R/_exp/post3/gpurun_out/r4/post3 $R/gpurun_out/r4/r/ 2>/dev/null
