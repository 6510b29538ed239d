"""BASELINE config E (die-reorient, 4096 envs on one MI355X, recurrent LSTM policy): env-steps/s of rollout +
PPO update.  Not the headline bench.  The env step is the step kernel's MYO_TASK_REORIENT task (csrc/myo_task.h); the LSTM
policy trains by hand-written BPTT replayed from a hipGraph."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--n-steps", type=int, default=32)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    import torch
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    from myochallenge_amd.rl.policy import ActorCriticPolicy
    from myochallenge_amd.rl.ppo import PPO, PPOConfig
    from myochallenge_amd.rl.vec_normalize import VecNormalize
    env = EnvironmentFactory.create("CustomMyoReorientP1", num_envs=a.envs, seed=1)
    torch.manual_seed(0)
    # src/main_reorient.py:53-71: LSTM-256 (actor + critic) -> [256, 256] ReLU, log_std_init -2
    pol = ActorCriticPolicy(env.obs_dim, env.act_dim, (256, 256), (256, 256), lstm_hidden_size=256, log_std_init=-2.0)
    algo = PPO(VecNormalize(env), pol, PPOConfig(n_steps=a.n_steps, batch_size=a.envs * a.n_steps // 8, n_epochs=4, learning_rate=2.5e-5))
    algo.collect_rollouts(); algo.train()                       # warm-up
    torch.cuda.synchronize()
    t0 = time.time(); tr = 0.0
    for _ in range(a.iters):
        t1 = time.time(); algo.collect_rollouts(); torch.cuda.synchronize(); tr += time.time() - t1
        algo.train()
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = a.iters * a.envs * a.n_steps
    # physics alone
    act = torch.zeros((a.envs, 39), device=env.device)
    t2 = time.time()
    for _ in range(50):
        env.step_tensor(act)
    torch.cuda.synchronize()
    print(json.dumps({"config": "E: CustomMyoReorientP1, %d envs, LSTM-256 + MLP[256,256]" % a.envs,
                      "env_steps_per_sec_rollout_plus_update": steps / dt, "env_steps_per_sec_rollout_only": steps / tr,
                      "env_steps_per_sec_env_only": 50 * a.envs / (time.time() - t2), "n_steps": a.n_steps, "epochs": 4}))


if __name__ == "__main__":
    main()
