import ctypes as C, torch, numpy as np, sys, os
from myochallenge_amd import native
from myochallenge_amd.rl.fused_mlp import FusedPPOStep, flatten_parameters
from myochallenge_amd.rl.policy import ActorCriticPolicy
lib = native.load(os.path.abspath(sys.argv[1]))
dev = torch.device("cuda:0")
B = 16384
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
out = (C.c_ulonglong * 16)()
lib.L.myo_debug_mlp_prof(out)
t = np.array(out[:9], dtype=np.int64)
names = ["idx/logstd", "gather+XT", "layer1", "layer2", "head", "loss", "colsum+dOT", "dH2", "dH1"]
print("wall_clock64 ticks (100 MHz = 10 ns):")
for n, d in zip(names[0:], np.diff(t)): print("  %-12s %6d ticks = %.2f us" % (n, d, d / 100.0))
print("  total %.2f us" % ((t[8] - t[0]) / 100.0))
