import os, time, torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig
from myochallenge_amd.rl.vec_normalize import VecNormalize
env = EnvironmentFactory.create("CustomMyoBaodingBallsP1", num_envs=4096, seed=1)
venv = VecNormalize(env)
torch.manual_seed(0)
pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=None)
algo = PPO(venv, pol, PPOConfig(n_steps=64, batch_size=16384, n_epochs=int(os.environ.get("EPOCHS","10")), learning_rate=float(os.environ.get("LR","3e-4"))))
algo.collect_rollouts()
algo.train(); torch.cuda.synchronize()
vm=[]
for i in range(6):
    algo.rollout_step(); torch.cuda.synchronize(); vm.append(float(venv.obs_rms.var.min()))
print(os.environ.get("TAG"), vm, "raw obs absmax", float(env._obs.abs().max()), "act", float(algo._act_s.abs().max()))
