import time, torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig
from myochallenge_amd.rl.vec_normalize import VecNormalize
env = EnvironmentFactory.create("CustomMyoBaodingBallsP1", num_envs=4096, seed=1)
venv = VecNormalize(env)
pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=None)
algo = PPO(venv, pol, PPOConfig(n_steps=64, batch_size=16384, n_epochs=10))
def sync(): torch.cuda.synchronize()
algo.collect_rollouts(); sync()
t=time.time(); algo.train(); sync(); print("first train (capture)", time.time()-t)
t=time.time(); algo.train(); sync(); print("train", time.time()-t, "per opt step ms", (time.time()-t)/160*1e3)
g=algo._gs
t=time.time()
for _ in range(50): algo._graph_fb.replay()
sync(); print("graph_fb replay ms", (time.time()-t)/50*1e3)
t=time.time()
for _ in range(50): algo._graph_ap.replay()
sync(); print("graph_ap replay ms", (time.time()-t)/50*1e3)
t=time.time()
for _ in range(10): perm = torch.randperm(262144, generator=algo.gen).to("cuda")
sync(); print("randperm+h2d ms", (time.time()-t)/10*1e3)
t=time.time()
for _ in range(50): g["idx"].copy_(perm[:16384])
sync(); print("idx copy ms", (time.time()-t)/50*1e3)
from myochallenge_amd.rl.ppo import compute_gae
t=time.time()
for _ in range(5): compute_gae(algo.rew_buf, algo.val_buf, algo.start_buf, algo._last_values, algo._last_starts, 0.99, 0.95)
sync(); print("gae ms", (time.time()-t)/5*1e3)
t=time.time(); algo.collect_rollouts(); sync(); print("rollout 64 steps s", time.time()-t)
