# round 4, call s: njp_post3_kernel with the atomic list (round 3's scan kernel) and the gather-free seed bound
O=gpurun_out/r4/s; mkdir -p $O
python -m pytest tests/test_gpu_nj.py -x -q -m gpu > $O/test_gpu_nj.log 2>&1; rc=$?; echo "test_gpu_nj rc=$rc"; tail -3 $O/test_gpu_nj.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/test_gpu_nj.log | head -20; exit 1; fi
python -m pytest tests/test_gpu_natural_sizes.py tests/test_gpu_fullsize.py -x -q -m gpu -k "nj" > $O/test_natural_nj.log 2>&1; rc=$?; echo "natural + fullsize nj rc=$rc"; tail -2 $O/test_natural_nj.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/test_natural_nj.log | head -20; exit 1; fi
for v in "" "DPR_NJP_FLAGS=64" "DPR_NJP_SMALL=fused"; do
  echo "== 30k variant [$v]"; env $v python3 profiles/nj_target.py --no-torch --reps 3 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_30k.txt
done
for v in "" "DPR_NJP_SMALL=fused"; do
  echo "== 30k indel gaps [$v]"; env $v python3 profiles/nj_target.py --no-torch --reps 2 --indel-gaps 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_30k.txt
done
for v in "" "DPR_NJP_SMALL=fused"; do
  echo "== 100k variant [$v]"; env $v python3 profiles/nj_target.py --no-torch --tips 100000 --reps 2 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_100k.txt
done
bash profiles/prof.sh trace post3_list_30k python3 profiles/nj_target.py --reps 1 2>&1 | grep -E "^njp_|nj_ms"
for it in 3000 12000; do python3 profiles/nj_target.py --no-torch --phases $it > $O/phases_post3_list_$it.txt 2>&1; done
grep -v "^{" $O/phases_post3_list_3000.txt | tail -32
