#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT; mkdir -p gpurun_out/r03
L=myochallenge_amd/libmyobatch.so
for v in "--dtype f64" "--integrator rk4" "--env reorient" "--env p2 --envs 8192"; do
echo "== whole $v"; MYO_STEP_ORDER=0 MYO_STEP_SPLIT=0 timeout 300 python3 tools/dev/kab.py $L --rounds 1 $v 2>&1 | tail -1
echo "== split+LPT $v"; timeout 300 python3 tools/dev/kab.py $L --rounds 1 $v 2>&1 | tail -1
done
python3 bench.py --no-cpu-baseline > gpurun_out/r03/bench_split.json 2> gpurun_out/r03/bench_split.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench_split.json')); print(d['value'], d['ms_per_step'], d['env_kernel_ms'], d['ppo_optimizer_steps_per_sec']); print({k:(round(v['value']),v['env_kernel_ms']) for k,v in d['variants'].items()})"
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6
