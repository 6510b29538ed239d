"""Newton iteration statistics over a batch (f32 vs f64) after a short rollout."""
import numpy as np, torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
for dtype in ("f32", "f64"):
    env = EnvironmentFactory.create("CustomMyoBaodingBallsP1", num_envs=1024, seed=1, dtype=dtype)
    env.reset_tensor()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for _ in range(20):
        env.step_tensor(torch.clamp(torch.randn((1024, 39), device="cuda", generator=g) * 0.135, -1, 1))
    b = env.batch
    out = torch.zeros((1024, b.dump_size), dtype=torch.float64, device="cuda")
    ctrl = torch.full((1024, 39), 0.0759, dtype=torch.float64, device="cuda")
    b.forward_dump(ctrl, out); torch.cuda.synchronize()
    o = b.dump_offset("counts"); c = out[:, o:o + 4].cpu().numpy()
    print(dtype, "ncon mean", c[:, 0].mean(), "nefc mean", c[:, 1].mean(), "iter hist", np.bincount(c[:, 2].astype(int)), "mean", c[:, 2].mean())
