import time, torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig
from myochallenge_amd.rl.vec_normalize import VecNormalize
env = EnvironmentFactory.create("CustomMyoBaodingBallsP1", num_envs=4096, seed=1)
venv = VecNormalize(env)
torch.manual_seed(0)
pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=None)
algo = PPO(venv, pol, PPOConfig(n_steps=64, batch_size=16384, n_epochs=10))
env.batch.enable_timing(True)
for i in range(3):
    t=time.time(); algo.collect_rollouts(); torch.cuda.synchronize(); dt=time.time()-t
    print("rollout", i, dt/64*1e3, "ms/step; kernel ms", env.batch.kernel_ms(), "act absmax", float(algo.act_buf.abs().max()), "obs finite", bool(torch.isfinite(algo.obs_buf).all()), "dones/step", float(algo.start_buf.sum())/64, "val", float(algo.val_buf.abs().max()))
    t=time.time(); st=algo.train(); torch.cuda.synchronize(); print("train", time.time()-t, st)
