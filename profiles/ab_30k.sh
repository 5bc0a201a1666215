#!/bin/bash
# usage (on the GPU box): bash profiles/ab_30k.sh "ENV=.. ENV=.." ...   -> hot-path phase times of bench.py at 30 000 tips per environment
for envs in "$@"; do
  echo "== $envs"
  env $envs python3 bench.py --steps 3 --warmup 1 --no-cli --no-parity --no-cpu-baseline --no-other-configs --no-stream-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['hot_path']['phase_ms']['nj'], d['hot_path']['prune']['units_scanned'])"
done
