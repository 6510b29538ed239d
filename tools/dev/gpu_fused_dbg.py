import torch, copy
from myochallenge_amd import native
from myochallenge_amd.rl.fused_mlp import FusedPPOStep, flatten_parameters, ppo_mlp_step_grads
from myochallenge_amd.rl.policy import ActorCriticPolicy
torch.manual_seed(0)
dev = torch.device("cuda:0")
lib = native.load()
pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=None).to(dev)
pol2, pol3 = copy.deepcopy(pol), copy.deepcopy(pol)
B = 4096
obs = torch.randn(B, 86, device=dev)
with torch.no_grad():
    act = pol.act(obs, None, None)[0]
    oldlp = pol.evaluate_actions(obs, act)[1] + torch.randn(B, device=dev) * 0.05
adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)
s1 = FusedPPOStep(pol, lib, 0.2, 0.01, 0.7)
print("unmerged", s1.run(obs, act, oldlp, adv, ret))
flatten_parameters(pol2)
s2 = FusedPPOStep(pol2, lib, 0.2, 0.01, 0.7)
print("merged", s2.run(obs, act, oldlp, adv, ret))
print("fp32", ppo_mlp_step_grads(pol3, obs, act, oldlp, adv, ret, 0.2, 0.01, 0.7, bf16=False))
torch.cuda.synchronize()
for (n, p), q, r in zip(pol.named_parameters(), pol2.parameters(), pol3.parameters()):
    e = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
    print(f"{n:40s} unmerged-vs-fp32 {e(p.grad, r.grad):.4f}  merged-vs-fp32 {e(q.grad, r.grad):.4f}  merged-vs-unmerged {e(q.grad, p.grad):.4f}")
