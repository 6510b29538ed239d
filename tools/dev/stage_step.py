"""One small deterministic workload for the per-stage counters (tools/dev/stage_counts.sh): 4096 envs, fp64 (or mixed) stepper,
40 warm-up env steps, then 12; prints the state checksum.  argv: library [dtype] [integrator] [env: p1 | reorient]."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from myochallenge_amd import native  # noqa: E402
from myochallenge_amd.envs.config import make_task_cfg  # noqa: E402
from myochallenge_amd.model import compile_model  # noqa: E402
from myochallenge_amd.synth_hand import synthetic_hand, synthetic_hand_die  # noqa: E402

lib = native.load(os.path.abspath(sys.argv[1]))
dtype = native.MYO_MIXED if len(sys.argv) > 2 and sys.argv[2] == "mixed" else native.MYO_F64
integ = 1 if len(sys.argv) > 3 and sys.argv[3] == "rk4" else 0
dev = torch.device("cuda:0")
N = 4096
if len(sys.argv) > 4 and sys.argv[4] == "reorient":
    from myochallenge_amd.envs.reorient import make_reorient_cfg
    cm = compile_model(synthetic_hand_die(), integrator=integ, unsupported_contacts="drop")
    tc = make_reorient_cfg("CustomMyoReorientP1", cm)
else:
    cm = compile_model(synthetic_hand(), integrator=integ)
    tc = make_task_cfg("CustomMyoBaodingBallsP1", cm)
b = native.Batch(native.Model(cm, lib), tc, N, 0, 1, dtype)
obs = torch.zeros((N, b.obs_dim), dtype=torch.float32, device=dev)
rew = torch.zeros(N, dtype=torch.float32, device=dev)
done = torch.zeros(N, dtype=torch.uint8, device=dev)
b.reset(None, obs)
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.clamp(torch.randn((N, 39), device=dev, generator=g) * 0.135, -1, 1) for _ in range(16)]
for t in range(52):
    b.step(acts[t % 16], obs, rew, done)
qp = torch.zeros((N, cm.size("nq")), dtype=torch.float64, device=dev)
b.get_state(qp)
torch.cuda.synchronize()
print("checksum", hashlib.sha1(qp.cpu().numpy().tobytes()).hexdigest()[:12])
