# round 4, call l: the whole -m gpu suite as the driver runs it, then smoke()
O=gpurun_out/r4/l; mkdir -p $O
( time python -m pytest tests/ -x -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; tail -22 $O/pytest_gpu.log; tail -3 $O/pytest_time.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
