#!/bin/bash
# scan-grid sweep of the pruned NJ (GPU box): bash profiles/grid_sweep.sh "128 256 512 1024"
for g in $1; do
  export DPR_NJP_GRID=$g
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid', $g, 'nj %.1f ms units %d' % (d['phase_ms']['nj'], d['prune']['units_scanned']))"
done
