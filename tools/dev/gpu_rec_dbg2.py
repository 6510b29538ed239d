import sys, time, torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig, compute_gae
from myochallenge_amd.rl.vec_normalize import VecNormalize
N = 4096
env = EnvironmentFactory.create("CustomMyoReorientP1", num_envs=N, seed=1)
torch.manual_seed(0)
pol = ActorCriticPolicy(env.obs_dim, env.act_dim, (256, 256), (256, 256), lstm_hidden_size=256, log_std_init=-2.0)
algo = PPO(VecNormalize(env), pol, PPOConfig(n_steps=32, batch_size=N * 32 // 8, n_epochs=4, learning_rate=2.5e-5))
algo.collect_rollouts(); algo.train(); algo.collect_rollouts(); torch.cuda.synchronize()
tn = algo.start_buf[1:].nonzero(); print("mid starts", tn.tolist()[:5])
bad_envs = set(int(x[1]) for x in tn.tolist())
adv, ret = compute_gae(algo.rew_buf, algo.val_buf, algo.start_buf, algo._last_values, algo._last_starts, 0.99, 0.95)
g = algo._rg; m = 512
g["adv"].copy_(adv); g["ret"].copy_(ret)
for dst, src in zip(g["state0"], algo._rollout_state0): dst.copy_(src)
perm = torch.randperm(N, device="cuda")
snap = algo._flat_adam.snapshot()
for s in range(0, N, m):          # eager first
    g["idx"].copy_(perm[s:s + m]); algo._rec_forward_backward(); torch.cuda.synchronize()
    has = bool(set(perm[s:s + m].tolist()) & bad_envs)
    print("eager mb", s, has, "pl", float(g["pl"]), "grad finite", bool(torch.isfinite(algo._flat_grad).all()))
algo._flat_adam.restore(snap)
fl = pol._flat
slot = {n_: sl for p_, sl in zip(fl["params"], fl["slots"]) for n_, q_ in pol.named_parameters() if q_ is p_}
off, k = slot["lstm_actor.bias_ih_l0"]
def run(tag, graph):
    algo._flat_adam.restore(snap)
    g["idx"].copy_(perm[512:1024])
    (algo._rgraph_fb.replay() if graph else algo._rec_forward_backward()); torch.cuda.synchronize()
    gb = fl["g"][off:off + k].clone()
    bad = (~torch.isfinite(gb)).nonzero().flatten().tolist()
    print(tag, "nonfinite idx", bad, "values", [float(gb[i]) for i in bad], "absmax finite", float(gb[torch.isfinite(gb)].abs().max()))
    return gb
e = run("eager", False)
a1 = run("graph", True)
a2 = run("graph again", True)
bad = (~torch.isfinite(a1)).nonzero().flatten().tolist()
print("eager value at those idx", [float(e[i]) for i in bad], "max |graph-eager| elsewhere", float((a1 - e)[torch.isfinite(a1)].abs().max()))
sb = algo.start_buf.clone()
algo.start_buf[1:] = 0
run("graph, mid start removed", True)
algo.start_buf.copy_(sb)
algo.start_buf[5, int(perm[700])] = 1
run("graph, extra mid start", True)
