#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
export PYTHONPATH=$ROOT
cd $ROOT
python3 -m pytest tests/test_rl.py tests/test_gpu_parity.py -m gpu -q -x -k "fused or ppo or mlp or graph or rollout or vecnorm or sample or native" 2>&1 | tail -4
python3 bench.py --no-cpu-baseline --no-variants > $OUT/bench_adv.json 2> $OUT/bench_adv.err
python3 -c "
import json; d=json.load(open('$OUT/bench_adv.json')); print(d['value'], d['ms_per_step'], d['env_kernel_ms'], d['ppo_optimizer_steps_per_sec'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants > /tmp/stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats_mid.csv \;
