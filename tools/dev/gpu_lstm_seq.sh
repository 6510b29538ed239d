#!/bin/bash
# developer A/B: sequence kernels (csrc/myo_lstm_seq.h) against the per-time-step kernels — tests, then config E and the LSTM-128 shape both ways
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/lstm_seq; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lstm_seq or lstm_step" 2>&1 | tail -5
python -m pytest tests/test_reorient.py tests/test_gpu_parity.py -m gpu -q -k "lstm or recurrent" 2>&1 | tail -5
for s in 0 1; do
  MYO_LSTM_SEQ=$s python tools/bench_reorient.py --reference-settings --iters 4 > $O/cfgE_ref_seq$s.json 2>$O/err_$s.log; python - <<P
import json
d = json.loads(open("$O/cfgE_ref_seq$s.json").read().strip().splitlines()[-1])
print("seq=$s reference", round(d["env_steps_per_sec_rollout_plus_update"]), d["seconds_per_iteration_rollout_update"][-1])
P
  MYO_LSTM_SEQ=$s python tools/bench_reorient.py --iters 6 > $O/cfgE_light_seq$s.json 2>>$O/err_$s.log; python - <<P
import json
d = json.loads(open("$O/cfgE_light_seq$s.json").read().strip().splitlines()[-1])
print("seq=$s light", round(d["env_steps_per_sec_rollout_plus_update"]), d["seconds_per_iteration_rollout_update"][-1])
P
done
