#!/bin/bash
# Per-kernel rocprofv3 stats of the other modes (GPU box): bash profiles/run_profiles_aux.sh
# -> gpurun_out/r4/<tag>/<tag>_kernel_stats.csv through profiles/prof.sh
bash profiles/prof.sh stats place_m python3 profiles/place_bench.py 20000 2000 m | tail -1
bash profiles/prof.sh stats place_r python3 profiles/place_bench.py 20000 3000 r | tail -1
bash profiles/prof.sh stats dc_m python3 profiles/dc_bench.py 1000000 2000 m | tail -1 | cut -c1-200
bash profiles/prof.sh stats dc_r python3 profiles/dc_bench.py 100000 5000 r | tail -1 | cut -c1-200
bash profiles/prof.sh stats add python3 profiles/add_bench.py 100000 10000 1000 m | tail -1 | cut -c1-200
bash profiles/prof.sh stats exact python3 profiles/exact_bench.py 10000 | tail -1 | cut -c1-200
