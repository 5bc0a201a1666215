# round 4, call e: latency probe; the hardened row-sharded loop (tests + kernel times with 8 virtual ranks); streaming probe
O=gpurun_out/r4/e; mkdir -p $O
tools/bin/lat_probe > $O/lat_probe.json 2> $O/lat_probe.err; cat $O/lat_probe.json
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_multiproc.py -x -q -m gpu > $O/tests_sharded.log 2>&1; echo "sharded rc=$?"; tail -3 $O/tests_sharded.log
python -m pytest tests/test_gpu_nj.py -x -q -m gpu -k "stream" > $O/tests_nj_stream.log 2>&1; echo "nj stream rc=$?"; tail -2 $O/tests_nj_stream.log
bash profiles/prof.sh stats njs_mailbox_8_virtual_ranks_30k python3 profiles/njs_vworld_stats.py 30000 10000 256 8 2
bash profiles/prof.sh stats njs_peer_8_virtual_ranks_30k python3 profiles/njs_vworld_stats.py 30000 10000 256 8 1
bash profiles/prof.sh stats stream_30k python3 profiles/nj_target.py --mode stream --iters 400
