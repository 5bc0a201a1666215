#!/bin/bash
# A/B of two library builds on the same box: bash profiles/ab_100k.sh  (expects dipper_amd/libdipper_prev.so)
run() { python bench.py --tips 100000 --sites 2000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'dist %.0f nj %.0f ms units %d' % (d['phase_ms']['dist'], d['phase_ms']['nj'], d['prune']['units_scanned']))"; }
cp dipper_amd/libdipper_hip.so /tmp/new.so
run new
cp dipper_amd/libdipper_prev.so dipper_amd/libdipper_hip.so; run prev
cp /tmp/new.so dipper_amd/libdipper_hip.so; run new
cp dipper_amd/libdipper_prev.so dipper_amd/libdipper_hip.so; run prev
cp /tmp/new.so dipper_amd/libdipper_hip.so
