#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
export PYTHONPATH=$ROOT
cd $ROOT
python3 tools/dev/gpu_mlp_check.py 2>&1 | tail -6
python3 -m pytest tests/test_rl.py tests/test_gpu_parity.py -m gpu -q -x -k "fused or ppo or mlp or graph" 2>&1 | tail -4
python3 bench.py --no-cpu-baseline --no-variants > $OUT/bench_adv.json 2> $OUT/bench_adv.err
python3 -c "
import json; d=json.load(open('$OUT/bench_adv.json')); print(d['value'], d['ms_per_step'], d['env_kernel_ms'], d['ppo_optimizer_steps_per_sec'])"
