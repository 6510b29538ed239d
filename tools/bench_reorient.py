"""BASELINE config E (die-reorient, 4096 envs on one MI355X, recurrent LSTM policy): env-steps/s of rollout +
PPO update.  Not the headline bench (myochallenge_amd/rl/bench_reorient_lstm.py does the work; bench.py records the same
measurement as `variants.config_E_lstm256`)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--n-steps", type=int, default=32)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--dtype", default="f64", choices=["mixed", "f64"], help="f64 since round 6: the mixed stepper is not offered for the die (DESIGN.md §4)")
    ap.add_argument("--n-epochs", type=int, default=4)
    ap.add_argument("--reference-settings", action="store_true", help="n_steps 128, n_epochs 10, fp64 physics, the PPO settings of src/main_reorient.py:53-71")
    a = ap.parse_args()
    import torch  # noqa: F401  (before the library: one HIP runtime per process)
    from myochallenge_amd.rl.bench_reorient_lstm import run
    print(json.dumps(run(a.envs, a.n_steps, a.iters, dtype=a.dtype, n_epochs=a.n_epochs, reference_settings=a.reference_settings)))


if __name__ == "__main__":
    main()
