#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT; mkdir -p gpurun_out/r03
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout or bookkeeping or training_entry or gsde" 2>&1 | tail -5
python3 bench.py --no-cpu-baseline --no-variants > gpurun_out/r03/bench_roll.json 2> gpurun_out/r03/bench_roll.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench_roll.json')); print(d['value'], d['ms_per_step'], d['env_kernel_ms'], d['ppo_optimizer_steps_per_sec'])"
MYO_ROLLOUT_GEMM=1 python3 bench.py --no-cpu-baseline --no-variants > gpurun_out/r03/bench_roll0.json 2> gpurun_out/r03/bench_roll0.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench_roll0.json')); print('gemm path', d['value'], d['ms_per_step'], d['env_kernel_ms'], d['ppo_optimizer_steps_per_sec'])"
