cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python bench.py --no-cpu-baseline --no-variants --min-seconds 1.5 > gpurun_out/r03/bench_mlp.json 2> gpurun_out/r03/bench_mlp.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_mlp.json").read().strip().splitlines()[-1])
print("bench", round(d["value"]), d["ms_per_step"], d["env_kernel_ms"], d["ppo_optimizer_steps_per_sec"])
PY
python -m pytest tests/test_rl.py tests/test_gpu_parity.py -m gpu -x -q -k "ppo or fused or graph or adam or rank or train or rollout" 2>&1 | tail -8
