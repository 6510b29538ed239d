"""Train PPO on the batched Baoding env for a fixed number of env steps and evaluate before / after.

    python tools/train_demo.py --steps 30000000

An end-to-end check that the physics, the task layer and the PPO update pull in the same direction:
the deterministic evaluation return has to rise.  (Synthetic MyoHand-shaped model: the numbers are not
comparable with the reference's MyoSuite scores.)"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-name", default="CustomMyoBaodingBallsP1")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=30_000_000)
    ap.add_argument("--eval-episodes", type=int, default=512)
    ap.add_argument("--out", default=None)
    ap.add_argument("--dtype", default="f64", choices=["f64", "mixed"], help="stepper arithmetic (f64: the bench headline since round 4)")
    ap.add_argument("--lstm-hidden", type=int, default=0, help="recurrent policy: LSTM of this width for actor and critic in front of the "
                    "[256, 256] trunks (the reference's RecurrentPPO policy class); PPO settings of tools/bench_reorient.py")
    ap.add_argument("--reference-settings", action="store_true", help="recurrent policy with the PPO settings of src/main_reorient.py:53-71 "
                    "(n_steps 128, n_epochs 10, lr 2.55673e-5, clip 0.3, lambda 0.9 ...), 300-step episodes: BASELINE config E as the reference trains it")
    a = ap.parse_args()
    import torch
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    from myochallenge_amd.metrics.evaluation import evaluate_policy, summarize
    from myochallenge_amd.rl.policy import ActorCriticPolicy
    from myochallenge_amd.rl.ppo import PPO, PPOConfig
    from myochallenge_amd.rl.vec_normalize import VecNormalize
    cfgs = {"weighted_reward_keys": {"pos_dist_1": 5.0, "pos_dist_2": 5.0, "act_reg": 0.0, "alive": 1.0, "solved": 5.0,
                                     "done": 0.0, "sparse": 0.0}}
    if "Reorient" in a.env_name:      # the reward shaping of src/main_reorient.py:27-37
        cfgs = {"weighted_reward_keys": {"pos_dist": 0.5, "rot_dist": 0.02, "pos_dist_diff": 50, "rot_dist_diff": 5, "alive": 0.1,
                                         "act_reg": 0, "solved": 0.5, "done": 0, "sparse": 0}}
    if a.reference_settings:
        cfgs["max_episode_steps"] = 300
    env = EnvironmentFactory.create(a.env_name, num_envs=a.envs, seed=1, dtype=a.dtype, **cfgs)
    eval_env = EnvironmentFactory.create(a.env_name, num_envs=512, seed=999, dtype=a.dtype, **cfgs)
    venv = VecNormalize(env, gamma=0.99)
    torch.manual_seed(0)
    pol = ActorCriticPolicy(env.obs_dim, env.act_dim, (256, 256), (256, 256), lstm_hidden_size=a.lstm_hidden or None, log_std_init=-2.0)
    if a.reference_settings:
        cfg = PPOConfig(n_steps=128, batch_size=a.envs * 128 // 8, n_epochs=10, learning_rate=2.55673e-05, ent_coef=3.62109e-06, clip_range=0.3,
                        gamma=0.99, gae_lambda=0.9, max_grad_norm=0.7, vf_coef=0.835671, bf16=True)
    elif a.lstm_hidden:     # sequences = whole 32-step rollouts of an eighth of the envs per minibatch
        cfg = PPOConfig(n_steps=32, batch_size=a.envs * 32 // 8, n_epochs=4, learning_rate=2.5e-4, clip_range=0.2, ent_coef=2.5e-4,
                        vf_coef=0.5, gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, bf16=True)
    else:
        cfg = PPOConfig(n_steps=64, batch_size=16384, n_epochs=10, learning_rate=2.5e-4, clip_range=0.2,
                        ent_coef=2.5e-4, vf_coef=0.5, gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, bf16=True)
    algo = PPO(venv, pol, cfg)
    log = []

    def ev(tag):
        r = summarize(evaluate_policy(pol, eval_env, venv, a.eval_episodes, deterministic=True))
        r.update(tag=tag, timesteps=algo.num_timesteps)
        log.append(r)
        print(json.dumps(r), flush=True)
    ev("before")
    t0 = time.time()
    chunk = max(a.steps // 5, a.envs * 64)
    while algo.num_timesteps < a.steps:
        algo.learn(min(a.steps, algo.num_timesteps + chunk))
        torch.cuda.synchronize()
        ev("t=%.0fs" % (time.time() - t0))
    print("trained %d env steps in %.1f s (%.0f steps/s incl. evaluations)" % (algo.num_timesteps, time.time() - t0,
                                                                              algo.num_timesteps / (time.time() - t0)))
    # batch health: hand-off protocol errors, substeps that dropped contacts / limit rows beyond the scratch's capacity
    health = {"train": env.batch.health(), "eval": eval_env.batch.health()}
    print("batch health:", json.dumps(health))
    if any(v for h in health.values() for v in h.values()):
        print("WARNING: non-zero health counters — some substeps dropped constraint rows (see include/myobatch.h: myo_batch_health)")
    log.append({"health": health})
    if a.out:
        json.dump(log, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
