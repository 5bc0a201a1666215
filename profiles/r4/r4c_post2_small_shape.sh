set -o pipefail
mkdir -p gpurun_out/r4/post2s
O=gpurun_out/r4/post2s
python -m pytest tests/test_gpu_nj.py -x -q -m gpu > $O/test_gpu_nj.log 2>&1; echo "test_gpu_nj rc=$?" | tee -a $O/test_gpu_nj.log; tail -3 $O/test_gpu_nj.log
for v in "" "DPR_NJP_FLAGS=16" "DPR_NJP_FLAGS=32" "DPR_NJP_FLAGS=48" "DPR_NJP_POST2=0"; do
  echo "== variant [$v]"; env $v python3 profiles/nj_target.py --no-torch --reps 3 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_30k.txt
done
for v in "" "DPR_NJP_FLAGS=48" ; do
  echo "== 100k variant [$v]"; env $v python3 profiles/nj_target.py --no-torch --tips 100000 --reps 2 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/variants_100k.txt
done
