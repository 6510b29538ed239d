"""Deterministic evaluation of a trained agent — the batched counterpart of /root/reference/src/main_eval.py.

The reference script (src/main_eval.py:52-118) builds one env, loads ``VecNormalize`` statistics and a
``RecurrentPPO`` zip, and plays ``num_episodes`` deterministic episodes one after another, printing the
mean length / return with their standard errors every 10 episodes.  Same inputs and printed quantities
here; the episodes run in parallel on the GPU.

    python -m myochallenge_amd.main_eval --model trained_models/baoding_phase2/final.zip \
        --env-path trained_models/baoding_phase2/normalized_env_final.pkl --env-name CustomMyoBaodingBallsP2
"""
from __future__ import annotations

import argparse
import json

import numpy as np

# evaluation configuration of the reference script (src/main_eval.py:13-45)
DEFAULT_CONFIG = {
    "weighted_reward_keys": {"pos_dist_1": 0, "pos_dist_2": 0, "act_reg": 0, "alive": 0, "solved": 5, "done": 0, "sparse": 0},
    "enable_rsi": False, "rsi_probability": 0, "balls_overlap": False, "overlap_probability": 0,
    "noise_fingers": 0, "limit_init_angle": 3.141592653589793, "goal_time_period": [4, 6],
    "goal_xrange": (0.020, 0.030), "goal_yrange": (0.022, 0.032),
    "obj_size_range": (0.018, 0.024), "obj_mass_range": (0.030, 0.300), "obj_friction_change": (0.2, 0.001, 0.00002),
    "task_choice": "random",
}


def evaluate(model_path: str, env_path: str, env_name: str = "CustomMyoBaodingBallsP2", config: dict = None,
             num_episodes: int = 100, num_envs: int = 256, seed: int = 0, deterministic: bool = True, verbose: bool = True):
    from .envs.environment_factory import EnvironmentFactory
    from .metrics.evaluation import evaluate_policy, summarize
    from .rl.sb3_zip import load_policy
    from .rl.vec_normalize import VecNormalize
    config = dict(DEFAULT_CONFIG if config is None else config)
    env = EnvironmentFactory.create(env_name, num_envs=min(num_envs, num_episodes), seed=seed, **config)
    venv = VecNormalize.load(env_path, env)
    venv.training = False          # src/main_eval.py:66-67
    venv.norm_reward = False
    policy, _ = load_policy(model_path)
    policy.to(env.device)
    res = evaluate_policy(policy, env, venv, n_eval_episodes=num_episodes, deterministic=deterministic)
    out = summarize(res)
    if verbose:
        print(f"Average len: {out['mean_len']:.2f} +/- {out['len_err']:.2f}")
        print(f"Average rew: {out['mean_rew']:.2f} +/- {out['rew_err']:.2f}")
        print(f"\nFinished evaluating {model_path}!")
    return res, out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", required=True, help="stable-baselines3 model zip (PATH_TO_PRETRAINED_NET)")
    ap.add_argument("--env-path", required=True, help="VecNormalize pickle (PATH_TO_NORMALIZED_ENV)")
    ap.add_argument("--env-name", default="CustomMyoBaodingBallsP2")
    ap.add_argument("--num-episodes", type=int, default=100)
    ap.add_argument("--num-envs", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--config", default=None, help="JSON file with the env kwargs (default: the reference script's config)")
    ap.add_argument("--out", default=None, help="write per-episode returns / lengths to this .npz")
    a = ap.parse_args(argv)
    cfg = json.load(open(a.config)) if a.config else None
    res, _ = evaluate(a.model, a.env_path, a.env_name, cfg, a.num_episodes, a.num_envs, a.seed)
    if a.out:
        np.savez(a.out, **res)


if __name__ == "__main__":
    main()
