# round 4, call j: refined placement policy, small NJ shape knobs, row-sharded scan grids, start-up trace, CPU baseline validation
O=gpurun_out/r4/j; mkdir -p $O
echo "== place 100k unaligned, refined policy"; python3 profiles/place_bench.py 100000 3000 r 2>&1 | tail -1 | tee -a $O/place_policy.jsonl
for v in "" "DPR_PLACE_OVERLAP_ALWAYS=1"; do
  echo "== place 250k unaligned [$v]"; env $v python3 profiles/place_bench.py 250000 3000 r 2>&1 | tail -1 | tee -a $O/place_policy.jsonl
done
for v in "DPR_NJ_TG_SMALL=32" "DPR_NJ_TG_SMALL=128" "DPR_NJP_GRID=128" "DPR_NJP_GRID=512"; do
  echo "== 30k fused [$v]"; env $v python3 profiles/nj_target.py --no-torch --reps 2 2>&1 | grep -o '"nj_ms": [0-9.]*' | tee -a $O/nj_knobs_30k.txt
done
echo "== 30k, inherited indel gaps"; python3 profiles/nj_target.py --no-torch --reps 2 --indel-gaps 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/nj_indel_gaps_30k.txt
for g in 512 2048; do
  echo "== njs mailbox, 8 virtual ranks, scan grid $g"; DPR_NJS_GRID=$g bash profiles/prof.sh stats njs_mailbox_grid$g python3 profiles/njs_vworld_stats.py 30000 10000 256 8 2 2>&1 | grep -E "njs_scan|njs_post|us_per_iteration"
done
# start-up of the command: HIP API trace of one `dipper` run (the program directly behind --)
tools/bin/gen_synth --tips 30000 --sites 10000 --seed 1 --indel-gaps --fasta /dev/shm/r4j.fa
for k in 1 2 3; do dipper_amd/bin/dipper -i m -I /dev/shm/r4j.fa -O /dev/shm/r4j.nwk -m 2 -d 2 2>&1 | grep -E "Device ready|Input in|Tree Created" | tr '\n' ' '; echo; done | tee $O/cli_startup_plain.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/hiptrace -o t -- $GRAFT_REPO_ROOT/dipper_amd/bin/dipper -i m -I /dev/shm/r4j.fa -O /dev/shm/r4j.nwk -m 2 -d 2 > $GRAFT_REPO_ROOT/$O/hiptrace.out 2> $GRAFT_REPO_ROOT/$O/hiptrace.err )
S=$(find $O/hiptrace -name "t_hip_api_stats.csv" | head -1); [ -n "$S" ] && head -25 $S | tee $O/cli_hip_api_stats.csv
find $O/hiptrace -name "*_trace.csv" -size +20M -delete
rm -f /dev/shm/r4j.fa /dev/shm/r4j.nwk
echo "== cpu baseline validation"; python3 profiles/cpu_baseline_validation.py 30000 16 2> $O/cpu_baseline_validation.err | tee $O/cpu_baseline_validation.jsonl | tail -1
