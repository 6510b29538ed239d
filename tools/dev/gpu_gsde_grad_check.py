"""Developer check: fused recurrent minibatch step (a) vs autograd under bf16 autocast (b) vs autograd in float32 (c, the yardstick),
per-parameter gradient cosine / relative error — the lstm+mlp-gsde case of tests/test_reorient.py::test_fused_recurrent_step_matches_autograd."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig, compute_gae
from myochallenge_amd.rl.vec_normalize import VecNormalize
torch.manual_seed(0)
N, T, m = 128, 8, 64
arch, hidden, sde = (64,), 32, True
env = EnvironmentFactory.create("CustomMyoReorientP1", num_envs=N, seed=3)
pol = ActorCriticPolicy(env.obs_dim, env.act_dim, arch, arch, lstm_hidden_size=hidden, use_sde=sde)
with torch.no_grad():
    pol.log_std.fill_(-0.5); pol.log_std.add_(0.2 * torch.randn_like(pol.log_std))
pol2, pol3 = copy.deepcopy(pol), copy.deepcopy(pol)
mk = lambda p, **kw: PPO(VecNormalize(env), p, PPOConfig(n_steps=T, batch_size=T * m, n_epochs=1, ent_coef=0.01, **kw))
a = mk(pol)
os.environ["MYO_RECURRENT_AUTOGRAD"] = "1"
b = mk(pol2)
c = mk(pol3, bf16=False)
a.collect_rollouts(); a.collect_rollouts()
a.start_buf[3, ::5] = 1.0; a.start_buf[6, 1::7] = 1.0
for o in (b, c):
    for name in ("obs_buf", "act_buf", "rew_buf", "val_buf", "logp_buf", "start_buf"):
        getattr(o, name).copy_(getattr(a, name))
    o._rollout_state0 = tuple(x.clone() for x in a._rollout_state0)
adv, ret = compute_gae(a.rew_buf, a.val_buf, a.start_buf, a._last_values, a._last_starts, 0.99, 0.95)
idx = torch.randperm(N, device=a.device)[:m]
grads, losses = [], []
for algo in (a, b, c):
    g = algo._rec_stage(adv, ret, T, N, m)
    g["idx"].copy_(idx)
    algo._rec_forward_backward()
    torch.cuda.synchronize()
    losses.append((float(g["pl"]), float(g["vl"])))
    grads.append({n: p.grad.detach().float().clone() for n, p in algo.policy.named_parameters()})
print("losses fused / autocast / fp32:", losses)
def cr(x, y):
    return float((x * y).sum() / (x.norm() * y.norm() + 1e-30)), float((x - y).norm() / (y.norm() + 1e-30))
for n in grads[2]:
    print("%-40s |g| %.3e  fused-vs-autocast cos %.4f rel %.3f | fused-vs-fp32 cos %.4f rel %.3f | autocast-vs-fp32 cos %.4f rel %.3f" %
          ((n, float(grads[2][n].norm())) + cr(grads[0][n], grads[1][n]) + cr(grads[0][n], grads[2][n]) + cr(grads[1][n], grads[2][n])))
