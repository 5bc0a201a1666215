# round 4, call m: the Mash index build without rocPRIM (merge rounds + own scan) -- parity, then timings
O=gpurun_out/r4/m; mkdir -p $O
python -m pytest tests/test_gpu_mash_place.py tests/test_gpu_dc.py tests/test_gpu_cli.py tests/test_gpu_exact.py -x -q -m gpu > $O/tests_a.log 2>&1; rc=$?; echo "tests a rc=$rc"; tail -3 $O/tests_a.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests_a.log | head -20; exit 1; fi
python -m pytest tests/test_gpu_natural_sizes.py tests/test_gpu_fullsize.py tests/test_gpu_accuracy.py -x -q -m gpu -k "mash or config2 or unaligned or place" > $O/tests_b.log 2>&1; rc=$?; echo "tests b rc=$rc"; tail -3 $O/tests_b.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests_b.log | head -20; exit 1; fi
echo "== place 100k unaligned"; python3 profiles/place_bench.py 100000 3000 r 2>&1 | tail -1 | tee -a $O/timings.jsonl
echo "== add 500k + 50k through Mash"; python3 profiles/add_bench.py 500000 50000 3000 r 2>&1 | tail -1 | tee -a $O/timings.jsonl
bash profiles/prof.sh stats place_100k_unaligned python3 profiles/place_bench.py 100000 3000 r
