#!/bin/bash
# sweep of the NJ epoch parameters (GPU box): bash profiles/epoch_sweep.sh
for pct in 50 60 70 80; do for mn in 2048 512; do
  export DPR_NJ_EPOCH_PCT=$pct DPR_NJ_EPOCH_MIN=$mn
  python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pct',$pct,'min',$mn,'ms/step %.1f nj %.1f units %d'%(d['ms_per_step'],d['phase_ms']['nj'],d['prune']['units_scanned']))"
done; done
