"""Developer tool (GPU box): drift of the HIP steppers against the oracle over whole episodes, many action
streams at once (tests/parity_cases.episode_drift), and k_step time against the number of envs.
    python tools/dev/gpu_drift.py [nsteps] > gpurun_out/drift.log"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_cases as pc
from myochallenge_amd import native
from myochallenge_amd.synth_hand import synthetic_hand

lib = native.load()
mj = synthetic_hand()
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
streams = [(sg, seed) for sg in (0.08, 0.135) for seed in range(16)]
out = {}
for dt, name in ((native.MYO_MIXED, "mixed"), (native.MYO_F64, "f64")):
    t0 = time.time()
    r = pc.episode_drift(lib, mj, dt, streams, nsteps)
    mq, mo = r["err_qpos_rel"].max(1), r["err_obs_abs"].max(1)
    print(f"== {name}: {len(streams)} streams x {nsteps} env steps in {time.time()-t0:.1f} s; streams with max err <= 1e-4: "
          f"qpos {int((mq <= 1e-4).sum())}, obs {int((mo <= 1e-4).sum())}; median max qpos err {np.median(mq):.2e}")
    for e, st in enumerate(streams):
        print("  ", st, "ends", r["episode_ends"][e], "split", r["episode_end_disagreement_at"][e], "max q %.1e obs %.1e" % (mq[e], mo[e]),
              " q@20..", " ".join("%.0e" % v for v in r["err_qpos_rel"][e][19::20]))
    out[name] = {"streams": r["streams"], "max_q": mq.tolist(), "max_obs": mo.tolist(), "ends": r["episode_ends"],
                 "split": r["episode_end_disagreement_at"], "q_every_10": r["err_qpos_rel"][:, 9::10].tolist()}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "drift_streams.json"), "w"))

# k_step time against the number of envs
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
for dtype in ("f32", "f64"):
    row = []
    for N in (256, 1024, 2048, 4096, 8192):
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
        row.append((N, round((time.time() - t) / 30 * 1e3, 3)))
        env.close()
    print("k_step ms/launch", dtype, row)
