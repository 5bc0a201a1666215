#!/bin/bash
# Per-kernel rocprofv3 stats of the other modes (GPU box): bash profiles/run_profiles_aux.sh <tag>
# -> gpurun_out/prof_<tag>_{place_m,place_r,dc_m,dc_r,add,exact}/b_kernel_stats.csv
T=${1:-aux}
bash profiles/prof_any.sh ${T}_place_m profiles/place_bench.py 20000 2000 m | tail -1
bash profiles/prof_any.sh ${T}_place_r profiles/place_bench.py 20000 3000 r | tail -1
bash profiles/prof_any.sh ${T}_dc_m profiles/dc_bench.py 1000000 2000 m | tail -1 | cut -c1-200
bash profiles/prof_any.sh ${T}_dc_r profiles/dc_bench.py 100000 5000 r | tail -1 | cut -c1-200
bash profiles/prof_any.sh ${T}_add profiles/add_bench.py 100000 10000 1000 m | tail -1 | cut -c1-200
bash profiles/prof_any.sh ${T}_exact profiles/exact_bench.py 10000 | tail -1 | cut -c1-200
