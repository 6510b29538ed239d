"""Die-reorient training entry point — the batched counterpart of /root/reference/src/main_reorient.py.

Same flow and hyper-parameters (config :26-49, max_episode_steps 300 :51, model_config :53-71: LSTM policy with
[256, 256] ReLU heads), one batched GPU env instead of 16 SubprocVecEnv workers.  Differences: the evaluation
callback gets its own env (the reference evaluates on the training env, which disturbs the running rollout),
and the minibatch is scaled with the number of envs (the reference's 32 of 2048 samples = 1/64 of a rollout).

    python -m myochallenge_amd.main_reorient --num-envs 1024 --timesteps 2000000
"""
from __future__ import annotations

import argparse
import os
from datetime import datetime

import torch.nn as nn

ENV_NAME = "CustomMyoReorientP1"

config = {
    "weighted_reward_keys": {"pos_dist": 0.5, "rot_dist": 0.02, "pos_dist_diff": 50, "rot_dist_diff": 5, "alive": 0.1, "act_reg": 0,
                             "solved": 0.5, "done": 0, "sparse": 0},
    "goal_pos": (-0.02, 0.02), "goal_rot": (-3.14, 3.14),
    "obj_size_change": 0, "obj_friction_change": (0, 0, 0),
    "enable_rsi": True, "rsi_distance_pos": 0, "rsi_distance_rot": 0,
    "goal_rot_x": None, "goal_rot_y": None, "goal_rot_z": None,
}
max_episode_steps = 300  # default: 150

model_config = dict(
    device="cuda", batch_size=32, n_steps=128, learning_rate=2.55673e-05, ent_coef=3.62109e-06, clip_range=0.3, gamma=0.99,
    gae_lambda=0.9, max_grad_norm=0.7, vf_coef=0.835671, n_epochs=10,
    policy_kwargs=dict(ortho_init=False, log_std_init=-2, activation_fn=nn.ReLU, net_arch=[dict(pi=[256, 256], vf=[256, 256])]),
)


def make_parallel_envs(env_config, num_env, start_index=0, **batch_kw):
    from .envs.environment_factory import EnvironmentFactory
    return EnvironmentFactory.create(ENV_NAME, num_envs=num_env, seed=start_index, max_episode_steps=max_episode_steps,
                                     **batch_kw, **env_config)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--num-envs", type=int, default=1024)
    ap.add_argument("--timesteps", type=int, default=10_000_000)
    ap.add_argument("--log-dir", default=None)
    ap.add_argument("--load-model", default=None)
    ap.add_argument("--load-env", default=None)
    ap.add_argument("--eval-freq", type=int, default=1_000_000, help="env TIMESTEPS between evaluations; SB3\'s EvalCallback counts vec-env steps (the reference: 10_000 steps of its 16 workers), so the callback gets this // num_envs")
    ap.add_argument("--save-freq", type=int, default=2_500_000, help="env TIMESTEPS between checkpoints (SB3 counts vec-env steps: the callback gets this // num_envs)")
    a = ap.parse_args(argv)
    from .metrics import CheckpointCallback, EnvDumpCallback, EvalCallback, TensorboardCallback
    from .rl.vec_normalize import VecNormalize
    from .train.trainer import MyoTrainer
    log_dir = a.log_dir or os.path.join("output", "training", datetime.now().strftime("%Y-%m-%d/%H-%M-%S"))
    os.makedirs(log_dir, exist_ok=True)
    envs = make_parallel_envs(config, a.num_envs)
    envs = VecNormalize.load(a.load_env, envs) if a.load_env else VecNormalize(envs)
    eval_env = VecNormalize(make_parallel_envs(config, min(256, a.num_envs), start_index=4242))
    mc = dict(model_config)
    mc["batch_size"] = max(32, a.num_envs * mc["n_steps"] // 64)          # 32 of 16 x 128 in the reference
    eval_callback = EvalCallback(eval_env=eval_env, callback_on_new_best=EnvDumpCallback(log_dir, verbose=0), n_eval_episodes=64,
                                 best_model_save_path=log_dir, log_path=log_dir, eval_freq=max(1, a.eval_freq // a.num_envs), deterministic=True, verbose=1)
    checkpoint_callback = CheckpointCallback(save_freq=max(1, a.save_freq // a.num_envs), save_path=log_dir, save_vecnormalize=True, verbose=1)
    tensorboard_callback = TensorboardCallback(info_keywords=("pos_dist", "rot_dist", "pos_dist_diff", "rot_dist_diff", "act_reg",
                                                              "alive", "solved"))
    trainer = MyoTrainer(envs=envs, env_config=config, load_model_path=a.load_model, log_dir=log_dir, model_config=mc,
                         callbacks=[eval_callback, checkpoint_callback, tensorboard_callback], timesteps=a.timesteps)
    trainer.train(total_timesteps=trainer.timesteps)
    trainer.save()


if __name__ == "__main__":
    main()
