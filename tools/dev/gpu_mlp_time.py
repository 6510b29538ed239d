"""kernel times of the fused PPO step for a given library (torch profiler), B = 16384"""
import sys, os, torch
from myochallenge_amd import native
from myochallenge_amd.rl.fused_mlp import FusedPPOStep, flatten_parameters
from myochallenge_amd.rl.policy import ActorCriticPolicy
lib = native.load(os.path.abspath(sys.argv[1]))
dev = torch.device("cuda:0")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=None).to(dev)
obs = torch.randn(B, 86, device=dev)
with torch.no_grad():
    act = pol.act(obs, None, None)[0]
    oldlp = pol.evaluate_actions(obs, act)[1]
adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)
flatten_parameters(pol)
step = FusedPPOStep(pol, lib, 0.2, 0.01, 0.7)
for _ in range(5): step.run(obs, act, oldlp, adv, ret)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(20): step.run(obs, act, oldlp, adv, ret)
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
    print("%-60s n=%3d avg %.1f us" % (e.key[:60], e.count, e.device_time_total / e.count))
