"""gSDE MLP policy, BASELINE config B shape: seconds per PPO update on the fused step (hipGraph) against eager autograd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig
from myochallenge_amd.rl.vec_normalize import VecNormalize
for graphs in (True, False):
    env = EnvironmentFactory.create("CustomMyoBaodingBallsP1", num_envs=4096, seed=1, dtype="mixed")
    torch.manual_seed(0)
    pol = ActorCriticPolicy(env.obs_dim, env.act_dim, (256, 256), (256, 256), lstm_hidden_size=None, use_sde=True, log_std_init=-2.0)
    algo = PPO(VecNormalize(env), pol, PPOConfig(n_steps=32, batch_size=16384, n_epochs=4, use_graphs=graphs))
    algo.collect_rollouts(); algo.train()
    torch.cuda.synchronize()
    tr = tu = 0.0
    for _ in range(3):
        t0 = time.time(); algo.collect_rollouts(); torch.cuda.synchronize(); t1 = time.time(); algo.train(); torch.cuda.synchronize()
        tr += t1 - t0; tu += time.time() - t1
    n = 3 * 4096 * 32
    print(f"use_graphs={graphs}: fused={algo._fused is not None}  {n / (tr + tu):,.0f} env-steps/s with update, rollout {tr / 3 * 1e3:.1f} ms, update {tu / 3 * 1e3:.1f} ms ({32} minibatch steps)")
    env.close()
