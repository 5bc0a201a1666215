#!/bin/bash
# exact placement after the climbing top-tree pass: parity tests, then 30 000 tips with the phase clocks, then without
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_exact.py -x -q > gpurun_out/exact_tests.log 2>&1 || { tail -30 gpurun_out/exact_tests.log; exit 1; }
tail -3 gpurun_out/exact_tests.log
DPR_EXACT_CLOCKS=1 timeout -k 10 200 python3 profiles/exact_bench.py 30000 2000 > gpurun_out/exact_clocks.txt 2>&1
cat gpurun_out/exact_clocks.txt
timeout -k 10 200 python3 profiles/exact_bench.py 30000 2000 > gpurun_out/exact_plain.txt 2>&1
cat gpurun_out/exact_plain.txt
