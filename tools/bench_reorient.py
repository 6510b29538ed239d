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
    a = ap.parse_args()
    from myochallenge_amd.rl.bench_reorient_lstm import run
    print(json.dumps(run(a.envs, a.n_steps, a.iters)))


if __name__ == "__main__":
    main()
