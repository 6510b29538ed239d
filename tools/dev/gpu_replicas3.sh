#!/bin/bash
export MYO_DIST_BACKEND=gloo OMP_NUM_THREADS=2
run() { python bench.py --steps 8 --warmup 0 --n-epochs 2 --min-seconds 0 --no-variants --no-cpu-baseline --dtype f64 "$@" 2>&1 | grep -v Warning | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('MYO_DP'): print(ln.strip())
    if ln.startswith('{'):
        d = json.loads(ln); print('$*', 'identical', d.get('replicas_identical'), 'spread', d.get('replica_checksum_spread'))
"; }
for i in 1 2 3; do run --gpus 8 --envs 256; done
for i in 1 2 3; do run --gpus 4 --envs 1024; done
for i in 1 2; do MYO_DP_CHECK=1 run --gpus 8 --envs 256; done
