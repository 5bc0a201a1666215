# round 4, call i: the per-batch overlap policy of the placement loop -- parity, then configs[4] / configs[2] timings per policy
O=gpurun_out/r4/i; mkdir -p $O
python -m pytest tests/test_gpu_mash_place.py tests/test_gpu_natural_sizes.py -x -q -m gpu -k "place or mash or import" > $O/tests_place.log 2>&1; rc=$?; echo "place tests rc=$rc"; tail -3 $O/tests_place.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests_place.log | head -20; exit 1; fi
for v in "" "DPR_PLACE_OVERLAP_ALWAYS=1" "DPR_PLACE_NO_OVERLAP=1"; do
  echo "== add 500k + 50k through Mash [$v]"; env $v python3 profiles/add_bench.py 500000 50000 3000 r 2>&1 | tail -1 | tee -a $O/add_mash_policies.jsonl
done
echo "== add 500k + 50k aligned"; python3 profiles/add_bench.py 500000 50000 1000 m 2>&1 | tail -1 | tee -a $O/add_aligned.jsonl
for v in "" "DPR_PLACE_OVERLAP_ALWAYS=1" "DPR_PLACE_NO_OVERLAP=1"; do
  echo "== place 100k unaligned [$v]"; env $v python3 profiles/place_bench.py 100000 3000 r 2>&1 | tail -1 | tee -a $O/place100k_policies.jsonl
done
bash profiles/prof.sh stats add_mash_500k_plus_50k python3 profiles/add_bench.py 500000 50000 3000 r
