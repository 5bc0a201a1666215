#!/bin/bash
# epoch threshold sweep at 100 000 tips only (GPU box): bash profiles/epoch_sweep3.sh "75 84"
for p in $1; do
  export DPR_NJ_EPOCH_PCT=$p
  python bench.py --tips 100000 --sites 2000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('100k pct', $p, 'nj %.0f ms units %d' % (d['phase_ms']['nj'], d['prune']['units_scanned']))"
done
