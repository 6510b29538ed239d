"""Time the PPO update alone (graph replay) and list the kernels of one minibatch step."""
import sys, time, torch
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
t=time.time(); algo.train(); sync(); dt=time.time()-t; print("train", dt, "per opt step ms", dt/160*1e3)
t=time.time()
for _ in range(100): algo._graph_fb.replay()
sync(); print("graph_fb replay ms", (time.time()-t)/100*1e3)
if len(sys.argv) > 1:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5): algo._graph_fb.replay()
        sync()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
t=time.time(); algo.collect_rollouts(); sync(); print("rollout 64 steps s", time.time()-t)
