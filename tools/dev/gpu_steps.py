"""Runs a handful of k_step launches (no PPO) — target for rocprofv3 --pmc passes."""
import sys, torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
env = EnvironmentFactory.create("CustomMyoBaodingBallsP1", num_envs=N, seed=1, dtype=dtype)
env.reset_tensor()
for _ in range(8):
    env.step_tensor(torch.clamp(torch.randn((N, 39), device="cuda") * 0.135, -1, 1))
torch.cuda.synchronize()
