"""Two physics-only batches (no task layer, no resets) stepped side by side from the same state with the same controls; the first
step after which they differ (developer tool).  python tools/dev/soak_physics.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from myochallenge_amd import native  # noqa: E402
from myochallenge_amd.model import compile_model  # noqa: E402
from myochallenge_amd.synth_hand import synthetic_hand  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 10
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # compare every N launches
lib = native.load(os.path.abspath(os.environ['SOAK_LIB'])) if os.environ.get('SOAK_LIB') else native.load()
dev = torch.device("cuda:0")
N = 4096
mj = synthetic_hand()
if os.environ.get("SOAK_NO_DAMPING"):
    mj.arrays["dof_damping"][:] = 0      # no implicit-in-damping solve in the Euler step
cm = compile_model(mj)
nq, nv, na = cm.size("nq"), cm.size("nv"), cm.size("na")
g = torch.Generator(device="cuda"); g.manual_seed(0)
ctrls = [torch.rand((N, 39), device=dev, generator=g, dtype=torch.float64) * 0.6 for _ in range(16)]
q0 = torch.tensor(cm.fields["qpos0"] if "qpos0" in cm.fields else [0.0] * nq, dtype=torch.float64, device=dev).repeat(N, 1) if False else None


def make():
    return native.Batch(native.Model(cm, lib), None, N, 0, 1, native.MYO_F64)


def state(b):
    q = torch.zeros((N, nq), dtype=torch.float64, device=dev); v = torch.zeros((N, nv), dtype=torch.float64, device=dev)
    a = torch.zeros((N, na), dtype=torch.float64, device=dev); t = torch.zeros(N, dtype=torch.float64, device=dev)
    b.get_state(q, v, a, t)
    return q, v, a


A, B = make(), make()
# the hand's wrist down (as the task's reset does) so that the balls rest in the palm
q, v, a = state(A)
q[:, 0] = -1.57
zero_t = torch.zeros(N, dtype=torch.float64, device=dev)
A.set_state(q, v, a, zero_t); B.set_state(q, v, a, zero_t)
found = 0
for t in range(steps):
    A.physics_step(ctrls[(t * nsub // 10) % 16], nsub)
    B.physics_step(ctrls[(t * nsub // 10) % 16], nsub)
    if (t * nsub) % 2000 == 2000 - nsub:      # every 200 steps: back to the start pose (the balls roll off the hand otherwise), velocities kept out
        A.set_state(q, v, a, zero_t); B.set_state(q, v, a, zero_t)
    if t % every != every - 1 and not found:
        continue
    sa, sb = state(A), state(B)
    ne = [(x != y) & ~(torch.isnan(x) & torch.isnan(y)) for x, y in zip(sa, sb)]
    if any(bool(d.any()) for d in ne):
        envs = torch.nonzero(torch.stack([d.any(1) for d in ne]).any(0)).flatten().tolist()
        e = envs[0]
        print(f"step {t}: {len(envs)} env(s) differ: {envs[:8]}; env {e} max |dq| {float((sa[0][e] - sb[0][e]).abs().max()):.3e} |dv| {float((sa[1][e] - sb[1][e]).abs().max()):.3e}")
        found += 1
        if found >= 2:
            break
print("health (slot check build: [0] slot found occupied, [1] workgroup ended in another slot than it started in, [2] map changed) A", A.health(), "B", B.health())
print("done: %d steps, %s; NaN envs %d" % (t + 1, "no difference" if not found else "differences above", int(torch.isnan(sa[0]).any(1).sum())))
