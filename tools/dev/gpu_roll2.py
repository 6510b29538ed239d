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
def chk(tag):
    torch.cuda.synchronize()
    bad = [n for n,p in pol.named_parameters() if not torch.isfinite(p).all()]
    print(tag, "params bad:", bad, "obs_s finite", bool(torch.isfinite(algo._obs_s).all()), "rms", bool(torch.isfinite(venv.obs_rms.mean).all()), bool(torch.isfinite(venv.obs_rms.var).all()), float(venv.obs_rms.count), "ret var", float(venv.ret_rms.var), "VARMIN", float(venv.obs_rms.var.min()), int(venv.obs_rms.var.argmin()), "raw obs finite", bool(torch.isfinite(env._obs).all()))
algo.collect_rollouts(); chk("after rollout0")
algo.collect_rollouts(); chk("after rollout1 (no train)")
algo.train(); chk("after train")
for i in range(3):
    algo.rollout_step(); chk(f"step {i}")
    print("  act", float(algo._act_s.abs().max()), "val", float(algo._val_s.abs().max()), "clip", float(algo._clip_s.abs().max()))
bad = ~torch.isfinite(algo._obs_s)
print("bad entries", int(bad.sum()), "rows", bad.any(1).nonzero().flatten()[:10].tolist(), "cols", bad.any(0).nonzero().flatten()[:20].tolist())
e = venv.normalize_obs(env._obs)
print("eager normalize finite", bool(torch.isfinite(e).all()), "diff rows", int((~torch.isfinite(algo._obs_s)).any(1).sum()))
r = bad.any(1).nonzero().flatten()[0]
print("raw row", env._obs[r, :8], "norm row", algo._obs_s[r, :8], "eager", e[r, :8])
print("var min", float(venv.obs_rms.var.min()), "old_obs finite", bool(torch.isfinite(venv.old_obs).all()))
