#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT; mkdir -p gpurun_out/r03
timeout 1200 python3 -m pytest tests/test_step_parts.py -m gpu -x -q 2>&1 | tail -4
python3 bench.py --no-cpu-baseline --no-variants > gpurun_out/r03/bench_split.json 2> gpurun_out/r03/bench_split.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench_split.json')); print(d['value'], d['ms_per_step'], d['env_kernel_ms'], d['ppo_optimizer_steps_per_sec'])"
