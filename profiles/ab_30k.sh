#!/bin/bash
# A/B of two library builds on the same box at the default size: bash profiles/ab_30k.sh (expects dipper_amd/libdipper_prev.so)
run() { python bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'nj %.1f ms' % d['phase_ms']['nj'])"; }
cp dipper_amd/libdipper_hip.so /tmp/new.so
for i in 1 2 3; do
  cp /tmp/new.so dipper_amd/libdipper_hip.so; run new
  cp dipper_amd/libdipper_prev.so dipper_amd/libdipper_hip.so; run prev
done
cp /tmp/new.so dipper_amd/libdipper_hip.so
