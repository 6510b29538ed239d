"""Occupancy experiment: time k_step with extra (unused) dynamic LDS per workgroup (MYO_LDS_PAD)."""
import os, sys, time, torch
from myochallenge_amd import native
native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmyobatch_pad.so")
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = EnvironmentFactory.create("CustomMyoBaodingBallsP1", num_envs=N, seed=1, dtype="f32")
env.reset_tensor()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.clamp(torch.randn((N, 39), device="cuda", generator=g) * 0.135, -1, 1) for _ in range(40)]
for a in acts[:10]:
    env.step_tensor(a)
torch.cuda.synchronize(); t = time.time()
for a in acts[10:]:
    env.step_tensor(a)
torch.cuda.synchronize()
print("pad", os.environ.get("MYO_LDS_PAD", "0"), "N", N, "ms/step", (time.time() - t) / 30 * 1e3)
