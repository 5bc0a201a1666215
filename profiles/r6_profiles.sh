# round 6 kernel statistics (rocprofv3 --kernel-trace --stats), one gpurun call:  bash profiles/r6_profiles.sh
export DPR_ROUND=r6
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6
bash profiles/prof.sh stats bench_hot_path_30k python3 profiles/nj_target.py --model gtr+g+i --indel-gaps --reps 2 && \
bash profiles/prof.sh stats dc_1m_10000_sites python3 profiles/protocol_dc.py 1000000 10000 && \
bash profiles/prof.sh stats place_100k_unaligned_10000_bases python3 profiles/place_bench.py 100000 10000 r && \
bash profiles/prof.sh stats exact_30k python3 profiles/exact_bench.py 30000 2000 && \
python3 profiles/protocol_add_setup.py /dev/shm/padd 500000 50000 10000 && \
DPR_CLI_NORMAL_EXIT=1 bash profiles/prof.sh stats add_50k_onto_500k_aligned_10000_sites dipper_amd/bin/dipper -i m -d 2 -a -t /dev/shm/padd/bb.nwk -I /dev/shm/padd/all.fa -O /dev/shm/padd/out.nwk
rc=$?
rm -rf /dev/shm/padd
for t in bench_hot_path_30k dc_1m_10000_sites place_100k_unaligned_10000_bases exact_30k add_50k_onto_500k_aligned_10000_sites; do echo "== $t"; tail -2 $O/$t/out.txt | cut -c1-600; tail -6 $O/$t/err.txt | grep -v rocprofv3 | cut -c1-300; done
exit $rc
