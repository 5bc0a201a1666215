# round 4, call t: exact float-reciprocal band in the device keys (nj_key_a_dev) -- parity on every NJ path, then timings
O=gpurun_out/r4/t; mkdir -p $O
python -m pytest tests/test_gpu_nj.py tests/test_gpu_sharded.py tests/test_gpu_multiproc.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head -20; exit 1; fi
python -m pytest tests/test_gpu_natural_sizes.py -x -q -m gpu -k "nj" > $O/natural.log 2>&1; echo "natural rc=$?"; tail -2 $O/natural.log
echo "== 30k"; python3 profiles/nj_target.py --no-torch --reps 3 2>&1 | grep -o '"nj_ms": [0-9.]*\|"digest": "[0-9a-f]*"' | paste - - | tee -a $O/timings.txt
echo "== 30k stream 400 iterations"; python3 profiles/nj_target.py --no-torch --mode stream --iters 400 --reps 2 2>&1 | grep -o '"nj_ms": [0-9.]*' | tee -a $O/timings.txt
echo "== 100k"; python3 profiles/nj_target.py --no-torch --tips 100000 --reps 2 2>&1 | grep -o '"nj_ms": [0-9.]*\|"digest": "[0-9a-f]*"' | paste - - | tee -a $O/timings.txt
bash profiles/prof.sh stats njs_mailbox_8vr_fastband python3 profiles/njs_vworld_stats.py 30000 10000 256 8 2 2>&1 | grep -E "njs_scan|njs_post"
