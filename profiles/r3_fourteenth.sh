#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
export DPR_NJ_NOGRAPH=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_p2 -o p2 --output-format csv -- python3 profiles/nj_kt.py 100000 10000 20000 1000 > $OUT/prof_p2.log 2>&1
echo "rc=$?"
tail -2 $OUT/prof_p2.log | cut -c1-300
f=$(find $OUT/prof_p2 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats_nj100k_first20k_post2.csv && head -8 $f | cut -c1-220
find $OUT/prof_p2 -name "*kernel_trace.csv" -delete
exit 0
