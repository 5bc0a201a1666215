# round 4, call g: njp_post3_kernel + cells mode -- parity first, then timing against the fused kernel, then phase stamps
O=gpurun_out/r4/g; mkdir -p $O
python -m pytest tests/test_gpu_nj.py -x -q -m gpu > $O/test_gpu_nj.log 2>&1; rc=$?; echo "test_gpu_nj rc=$rc"; tail -3 $O/test_gpu_nj.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/test_gpu_nj.log | head -20; exit 1; fi
python -m pytest tests/test_gpu_natural_sizes.py -x -q -m gpu -k nj_10k > $O/test_natural_nj.log 2>&1; echo "natural nj rc=$?"; tail -2 $O/test_natural_nj.log
for v in "" "DPR_NJP_SMALL=fused" "DPR_NJP_FLAGS=1"; do
  echo "== 30k variant [$v]"; env $v python3 profiles/nj_target.py --no-torch --reps 3 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_30k.txt
done
echo "== 30k with 3 % gaps"; python3 profiles/nj_target.py --no-torch --reps 2 --gap-frac 0.03 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_30k.txt
echo "== 30k with 3 % gaps, fused"; DPR_NJP_SMALL=fused python3 profiles/nj_target.py --no-torch --reps 2 --gap-frac 0.03 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_30k.txt
for v in "" "DPR_NJP_SMALL=fused"; do
  echo "== 100k variant [$v]"; env $v python3 profiles/nj_target.py --no-torch --tips 100000 --reps 2 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_100k.txt
done
for it in 3000 12000; do python3 profiles/nj_target.py --no-torch --phases $it > $O/phases_post3_$it.txt 2>&1; done
grep -v "^{" $O/phases_post3_3000.txt
