"""kernel times of the fused PPO step for a given library (torch profiler), B = 16384.
    python tools/dev/gpu_mlp_time.py <lib.so> [B] [--shuffled]   --shuffled: minibatch rows drawn at random from a 262,144-row rollout
    (the bench's condition: the gather of k_mlp_fwdbwd then reads scattered 344-byte rows from a 90 MB buffer)"""
import sys, os, torch
from myochallenge_amd import native
from myochallenge_amd.rl.fused_mlp import FusedPPOStep, flatten_parameters
from myochallenge_amd.rl.policy import ActorCriticPolicy
lib = native.load(os.path.abspath(sys.argv[1]))
dev = torch.device("cuda:0")
args = [a for a in sys.argv[2:] if not a.startswith('--')]
B = int(args[0]) if args else 16384
shuffled = '--shuffled' in sys.argv
pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=None).to(dev)
obs = torch.randn(B, 86, device=dev)
with torch.no_grad():
    act = pol.act(obs, None, None)[0]
    oldlp = pol.evaluate_actions(obs, act)[1]
adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)
flatten_parameters(pol)
step = FusedPPOStep(pol, lib, 0.2, 0.01, 0.7)
if shuffled:
    R = 262144
    g = torch.Generator(device=dev); g.manual_seed(0)
    obs_all = torch.randn(R, 86, device=dev); rows = torch.randint(0, B, (R,), device=dev, generator=g)
    act_all, oldlp_all = act[rows].contiguous(), oldlp[rows].contiguous()
    with torch.no_grad():
        oldlp_all = pol.evaluate_actions(obs_all[:B], act_all[:B])[1].repeat(R // B)
    adv_all, ret_all = torch.randn(R, device=dev), torch.randn(R, device=dev)
    idx = torch.randperm(R, device=dev, generator=g)[:B].contiguous()
    run = lambda: step.run_indexed(obs_all, act_all, oldlp_all, adv_all, ret_all, idx)
else:
    run = lambda: step.run(obs, act, oldlp, adv, ret)
for _ in range(5): run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(20): run()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
    print("%-60s n=%3d avg %.1f us" % (e.key[:60], e.count, e.device_time_total / e.count))
