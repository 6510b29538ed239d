"""A/B timing of k_step alone: python gpu_occ.py N dtype [libpath]  (MYO_LDS_PAD with a -DMYO_LDS_PAD_EXPERIMENT build)."""
import os, sys, time, torch
from myochallenge_amd import native
if len(sys.argv) > 3:
    native.LIB_PATH = os.path.abspath(sys.argv[3])
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
env = EnvironmentFactory.create("CustomMyoBaodingBallsP1", num_envs=N, seed=1, dtype=dtype)
env.reset_tensor()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.clamp(torch.randn((N, 39), device="cuda", generator=g) * 0.135, -1, 1) for _ in range(40)]
for a in acts[:10]:
    env.step_tensor(a)
torch.cuda.synchronize(); t = time.time()
for a in acts[10:]:
    env.step_tensor(a)
torch.cuda.synchronize()
print(os.path.basename(native.LIB_PATH), "pad", os.environ.get("MYO_LDS_PAD", "0"), "N", N, dtype, "ms/step", round((time.time() - t) / 30 * 1e3, 4))
