#!/bin/bash
# epoch threshold sweep at two sizes (GPU box): bash profiles/epoch_sweep2.sh "80 88" 
for p in $1; do
  export DPR_NJ_EPOCH_PCT=$p
  python bench.py --tips 100000 --sites 2000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('100k pct', $p, 'nj %.0f ms units %d' % (d['phase_ms']['nj'], d['prune']['units_scanned']))"
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(' 30k pct', $p, 'nj %.1f ms units %d' % (d['phase_ms']['nj'], d['prune']['units_scanned']))"
done
