#!/bin/bash
# developer diagnostic: replica check of bench.py --gpus N (gloo ranks sharing the one GPU) for several (N, envs, env) combinations
export MYO_DIST_BACKEND=gloo OMP_NUM_THREADS=2
for cfg in "2 256 CustomMyoBaodingBallsP1" "2 4096 CustomMyoBaodingBallsP1" "4 1024 CustomMyoBaodingBallsP1" "8 256 CustomMyoBaodingBallsP1" "2 4096 CustomMyoBaodingBallsP2"; do
  set -- $cfg
  python bench.py --gpus $1 --envs $2 --env-name $3 --steps 8 --warmup 0 --n-epochs 2 --min-seconds 0 --no-variants --no-cpu-baseline --dtype f64 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$cfg', 'identical', d.get('replicas_identical'), 'spread', d.get('replica_checksum_spread'), 'ppo', d['config']['ppo'], 'value', round(d['value']))
"
done
