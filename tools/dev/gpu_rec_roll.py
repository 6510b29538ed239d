"""Time the parts of the recurrent graphed rollout step (config E)."""
import sys, time, torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig
from myochallenge_amd.rl.vec_normalize import VecNormalize
N = 4096
env = EnvironmentFactory.create("CustomMyoReorientP1", num_envs=N, seed=1)
torch.manual_seed(0)
pol = ActorCriticPolicy(env.obs_dim, env.act_dim, (256, 256), (256, 256), lstm_hidden_size=256, log_std_init=-2.0)
algo = PPO(VecNormalize(env), pol, PPOConfig(n_steps=32, batch_size=N * 32 // 8, n_epochs=4, learning_rate=2.5e-5))
algo.collect_rollouts()
torch.cuda.synchronize()
def tm(f, n=20):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
print("gA", tm(algo._gA.replay), "env", tm(lambda: algo._raw.step_tensor(algo._clip_s)), "gB", tm(algo._gB.replay))
print("step", tm(algo.rollout_step))
print("phys", tm(lambda: env.batch.physics_step(env._ctrl, env.frame_skip, env._stream())))
t = time.time(); algo.train(); torch.cuda.synchronize(); print("train s", time.time() - t)
t = time.time(); algo.train(); torch.cuda.synchronize(); print("train s", time.time() - t)
print("after train: gA", tm(algo._gA.replay), "env", tm(lambda: algo._raw.step_tensor(algo._clip_s)), "gB", tm(algo._gB.replay))
print("step", tm(algo.rollout_step))
t = time.time(); algo.collect_rollouts(); torch.cuda.synchronize(); print("collect s", time.time() - t)
print("done frac", float(env._done.float().mean()), "finite", bool(torch.isfinite(env._obs).all()))
