"""Workgroup timeline of one k_step launch (-DMYO_WGTIME build): how much of slots x makespan is busy, spread of the
per-env durations, and the tail.  python tools/dev/gpu_wgtime.py <lib.so> [envs]"""
import ctypes as C, os, sys
import numpy as np, torch
from myochallenge_amd import native
from myochallenge_amd.envs.config import make_task_cfg
from myochallenge_amd.model import compile_model
from myochallenge_amd.synth_hand import synthetic_hand
lib = native.load(os.path.abspath(sys.argv[1]))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = int(sys.argv[3]) if len(sys.argv) > 3 else 4           # parts per env step of the library's plan (MYO_STEP_SPLIT)
dev = torch.device("cuda:0")
cm = compile_model(synthetic_hand(), integrator=0)
b = native.Batch(native.Model(cm, lib), make_task_cfg("CustomMyoBaodingBallsP1", cm), N, 0, 1, native.MYO_MIXED)
obs = torch.zeros((N, 86), dtype=torch.float32, device=dev); rew = torch.zeros(N, dtype=torch.float32, device=dev)
done = torch.zeros(N, dtype=torch.uint8, device=dev)
b.reset(None, obs)
g = torch.Generator(device="cuda"); g.manual_seed(0)
prev = None
for k in range(45):
    b.step(torch.clamp(torch.randn((N, 39), device=dev, generator=g) * 0.135, -1, 1), obs, rew, done)
    if k < 40: continue
    torch.cuda.synchronize()
    out = (C.c_ulonglong * (2 * N * P))()
    lib.L.myo_debug_read_wgtime(out, N * P)
    t = np.array(out[:], dtype=np.int64).reshape(N * P, 2)
    t0, t1 = t[:, 0] - t[:, 0].min(), t[:, 1] - t[:, 0].min()
    dur = (t1 - t0) / 100.0          # us
    span = t1.max() / 100.0
    slots = 2048
    print(f"step {k}: makespan {span:.0f} us; busy {dur.sum() / (slots * span):.3f} of {slots} slots; duration us: mean {dur.mean():.0f} "
          f"p5 {np.percentile(dur, 5):.0f} p50 {np.percentile(dur, 50):.0f} p95 {np.percentile(dur, 95):.0f} max {dur.max():.0f}; "
          f"last start {t0.max() / 100.0:.0f} us; WGs started in first 50 us: {(t0 < 5000).sum()}")
    r1 = t0 < 5000
    print(f"   first-round WGs: n {r1.sum()} mean {dur[r1].mean():.0f} us p95 {np.percentile(dur[r1], 95):.0f}; later: n {(~r1).sum()} mean {dur[~r1].mean():.0f} us p95 {np.percentile(dur[~r1], 95):.0f}")
    # occupancy over time
    edges = np.linspace(0, span, 11)
    occ = [int(((t0 / 100.0 <= e) & (t1 / 100.0 > e)).sum()) for e in edges[:-1] + span / 20]
    print("   running WGs at 5%,15%..95% of the makespan:", occ)
    if P > 1:
        for q in range(P): print(f"   part {q}: mean {dur[q * N:(q + 1) * N].mean():.0f} us, first start {t0[q * N:(q + 1) * N].min() / 100.0:.0f} us, last end {t1[q * N:(q + 1) * N].max() / 100.0:.0f} us")
        dur = dur.reshape(P, N).sum(0); prev = None if prev is None or prev.shape != dur.shape else prev
        continue
    dn = done.bool().cpu().numpy()
    if dn.any(): print(f"   envs that reset in this step: {dn.sum()}, mean duration {dur[dn].mean():.0f} us (others {dur[~dn].mean():.0f})")
    if k > 40: print(f"   correlation with the previous step's duration: {np.corrcoef(dur, prev)[0, 1]:.3f}; rel. error of 'same as last step' p50 {np.median(np.abs(dur - prev) / dur):.3f} p95 {np.percentile(np.abs(dur - prev) / dur, 95):.3f}")
    top = np.argsort(-dur)[:40]
    print(f"   40 longest: reset {dn[top].sum()}, mean {dur[top].mean():.0f} us; were in last step's top 200: {np.isin(top, np.argsort(-prev)[:200]).sum() if k > 40 else -1}")
    prev = dur.copy()
