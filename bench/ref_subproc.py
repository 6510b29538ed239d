#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path — SubprocVecEnv over CustomMyoBaodingBallsP2 workers — at 8 and 16 workers
(BASELINE.json configs[0]; /root/reference/src/main_baoding.py:56-65 builds the vec env, :74 uses 16 workers).

This cannot run in the build image or on the GPU box (MuJoCo 2.1 / MyoSuite 1.2.3 / gym 0.13 / SB3 1.6.2 are not
installed and the reference tree does not travel): it is the hook for a user who has the reference's
environment.  It imports the reference's EnvironmentFactory from --reference/src and measures env-steps/s of
`SubprocVecEnv.step` with N(0, e^-2) actions (SB3's initial policy noise), auto-resets included, no learner.

    python bench/ref_subproc.py --reference /path/to/myochallenge [--workers 8 16] [--steps 2000] [--env CustomMyoBaodingBallsP2]

Prints one JSON line per worker count: {"metric": "env-steps/sec", "value": ..., "workers": N, "cores": os.cpu_count(), ...}
(the shape bench.py's "cpu_baseline" object uses, with "kind": "reference")."""
import argparse
import json
import os
import sys
import time

import numpy as np

# the registration defaults of the P2 env are the reference's own; only the keys main_baoding.py:26-52 overrides
P2_CONFIG = {
    "weighted_reward_keys": {"pos_dist_1": 2, "pos_dist_2": 2, "act_reg": 0, "alive": 0, "solved": 5, "done": 0, "sparse": 0},
    "task_choice": "random", "enable_rsi": False, "rsi_probability": 0, "balls_overlap": False, "overlap_probability": 0,
    "noise_fingers": 0, "limit_init_angle": False, "goal_time_period": [4, 6], "goal_xrange": (0.020, 0.030),
    "goal_yrange": (0.022, 0.032), "obj_size_range": (0.018, 0.022), "obj_mass_range": (0.030, 0.300),
    "obj_friction_change": (0.2, 0.001, 0.00002),
}


def time_workers(env_name, config, workers, steps, warmup, seed):
    from stable_baselines3.common.monitor import Monitor
    from stable_baselines3.common.vec_env.subproc_vec_env import SubprocVecEnv
    from envs.environment_factory import EnvironmentFactory        # the reference's factory (src/envs/environment_factory.py)

    def thunk(_):
        return lambda: Monitor(EnvironmentFactory.create(env_name, **config))

    venv = SubprocVecEnv([thunk(i) for i in range(workers)])
    venv.seed(seed)
    venv.reset()
    rng = np.random.RandomState(seed)
    na = venv.action_space.shape[0]
    draw = lambda: np.clip(rng.normal(0, np.exp(-2.0), (workers, na)), -1, 1).astype(np.float32)
    for _ in range(warmup):
        venv.step(draw())
    t0 = time.perf_counter()
    for _ in range(steps):
        venv.step(draw())
    dt = time.perf_counter() - t0
    venv.close()
    return workers * steps / dt, dt


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="checkout of amathislab/myochallenge (its src/ is put on sys.path)")
    ap.add_argument("--workers", type=int, nargs="+", default=[8, 16])
    ap.add_argument("--steps", type=int, default=2000, help="vec-env steps per measurement")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--env", default="CustomMyoBaodingBallsP2")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    src = os.path.join(os.path.abspath(a.reference), "src")
    if not os.path.isdir(src):
        raise SystemExit(f"{src} not found")
    sys.path.insert(0, src)
    try:
        import envs  # noqa: F401  (registers the gym ids, src/envs/__init__.py)
    except Exception as exc:
        raise SystemExit(f"the reference environment cannot be imported here ({exc!r}): this script needs MuJoCo 2.1 + "
                         "MyoSuite 1.2.3 + gym 0.13 + stable-baselines3 1.6.2 as pinned in the reference's requirements.txt")
    config = P2_CONFIG if a.env.endswith("P2") else {}
    for w in a.workers:
        rate, dt = time_workers(a.env, config, w, a.steps, a.warmup, a.seed)
        print(json.dumps({"metric": "env-steps/sec", "value": rate, "unit": "env-steps/s", "workers": w, "cores": os.cpu_count(),
                          "kind": "reference", "env": a.env, "sample": f"{a.steps} SubprocVecEnv steps x {w} workers in {dt:.1f} s, "
                          "N(0, e^-2) actions, auto-reset, no learner"}))


if __name__ == "__main__":
    main()
