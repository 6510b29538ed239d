#!/bin/bash
export MYO_DIST_BACKEND=gloo OMP_NUM_THREADS=2 MYO_DP_CHECK=1
for i in 1 2 3; do python bench.py --gpus 8 --envs 256 --steps 8 --warmup 0 --n-epochs 2 --min-seconds 0 --no-variants --no-cpu-baseline --dtype f64 2>&1 | grep "MYO_DP" | head -8; echo "--"; done
