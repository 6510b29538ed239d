"""One batch: every env step, the forward-dynamics dump of the SAME state is taken twice; the first step at which the two dumps differ,
and in which field (developer tool; a nondeterministic stage of the fp64 stepper located).  python tools/dev/soak_dump.py [steps] [lib]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from myochallenge_amd import native  # noqa: E402
from myochallenge_amd.envs.config import make_task_cfg  # noqa: E402
from myochallenge_amd.model import compile_model  # noqa: E402
from myochallenge_amd.synth_hand import synthetic_hand  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
lib = native.load(os.path.abspath(sys.argv[2])) if len(sys.argv) > 2 else native.load()
dev = torch.device("cuda:0")
N = 4096
cm = compile_model(synthetic_hand())
tc = make_task_cfg("CustomMyoBaodingBallsP2", cm)
b = native.Batch(native.Model(cm, lib), tc, N, 0, 1, native.MYO_F64)
obs = torch.zeros((N, b.obs_dim), dtype=torch.float32, device=dev)
b.reset(None, obs)
rew = torch.zeros(N, dtype=torch.float32, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.clamp(torch.randn((N, 39), device=dev, generator=g) * 0.135, -1, 1) for _ in range(16)]
D = b.dump_size
names = ["ten_length", "ten_J", "M", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "qacc", "actuator_force", "act_dot", "counts",
         "efc_aref", "efc_D", "site_xpos", "subtree_com", "xpos"]
offs = sorted((b.dump_offset(n), n) for n in names) + [(D, "end")]
d1 = torch.zeros((N, D), dtype=torch.float64, device=dev); d2 = torch.zeros((N, D), dtype=torch.float64, device=dev)
ctrl = torch.full((N, 39), 0.3, dtype=torch.float64, device=dev)
found = 0
every = int(os.environ.get("SOAK_STEP_EVERY", "1"))      # one env step every N dump pairs (N > 1: mostly dumps)
for t in range(steps):
    if t % every == 0:
        b.step(acts[(t // every) % 16], obs, rew, done)
    b.forward_dump(ctrl, d1)
    b.forward_dump(ctrl, d2)
    ne = (d1 != d2) & ~(torch.isnan(d1) & torch.isnan(d2))
    if bool(ne.any()):
        envs = torch.nonzero(ne.any(1)).flatten().tolist()
        print(f"step {t}: dumps of the same state differ in {len(envs)} env(s): {envs[:8]}")
        e = envs[0]
        for (o, n), (o2, _) in zip(offs[:-1], offs[1:]):
            seg = ne[e, o:o2]
            if bool(seg.any()):
                k = int(torch.nonzero(seg).flatten()[0])
                print(f"   env {e} {n}: {int(seg.sum())} entries differ; [{k}] {float(d1[e, o + k]):.17g} vs {float(d2[e, o + k]):.17g}")
        found += 1
        if found >= 3:
            break
print("done: %d steps, %s" % (t + 1, "dumps always equal" if not found else "differences above"))
