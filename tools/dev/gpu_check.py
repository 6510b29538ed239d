"""Developer check on a real MI355X: HIP kernels vs the oracle + first timings.
Run through gpurun; prints everything to stdout."""
import sys
import time

import numpy as np
import torch

np.set_printoptions(precision=6, linewidth=180, suppress=True)
from myochallenge_amd import native
from myochallenge_amd.envs.config import make_task_cfg, task_ids
from myochallenge_amd.mjb import load_mjb
from myochallenge_amd.model import compile_model
from myochallenge_amd.synth_hand import synthetic_hand
from oracle.oracle import BaodingState, OracleData, OracleModel, baoding_step, make_cfg

dev = torch.device("cuda:0")
lib = native.load()
print(lib.version, torch.cuda.get_device_name(0))
rng = np.random.RandomState(0)


def T(a, dt=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)


def fwd_compare(name, mj, q=None, v=None, act=None, ctrl=None, dtype=native.MYO_F64):
    cm = compile_model(mj, unsupported_contacts="drop")
    om = OracleModel(cm.to_blob()); d = OracleData(om)
    nm = native.Model(cm, lib); b = native.Batch(nm, None, 4, 0, 0, dtype)
    nq, nv, na, nu = om.nq, om.nv, om.na, om.nu
    if q is not None: d.qpos[:] = q
    if v is not None: d.qvel[:] = v
    if act is not None: d.act[:] = act
    c = np.zeros((1, nu)) if ctrl is None else np.array(ctrl, float).reshape(1, nu)
    d.ctrl[:] = c[0]
    b.set_state(T(np.tile(d.qpos, (4, 1))), T(np.tile(d.qvel, (4, 1))), T(np.tile(d.act, (4, 1))), T(np.zeros(4)))
    out = torch.zeros((4, b.dump_size), dtype=torch.float64, device=dev)
    b.forward_dump(T(np.tile(c, (4, 1))), out)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    d.forward()
    res = {}
    for n, ref in [("ten_length", d.ten_length), ("ten_J", d.ten_J), ("M", d.M), ("qfrc_bias", d.qfrc_bias),
                   ("qfrc_actuator", d.qfrc_actuator), ("qacc_smooth", d.qacc_smooth), ("qacc", d.qacc)]:
        ref = np.array(ref); o = b.dump_offset(n); got = out[3, o:o + ref.size]
        res[n] = float("%.2e" % (np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)))
    o = b.dump_offset("counts")
    print(name, "counts gpu", out[3, o:o + 4], "oracle", d.ncon, d.nefc, d.solver_iter, d.nl, res)


mjf = load_mjb("tests/golden/myo_finger_v0.mjb")
fwd_compare("finger limits", mjf, q=np.array([0.5, 1.2, 1.2, 1.1]), v=rng.normal(0, 1, 4), act=rng.uniform(0, 1, 5), ctrl=rng.uniform(0, 1, 5))
mj = synthetic_hand()
q = mj.qpos0.copy(); q[0] = -1.57
fwd_compare("hand init", mj, q=q)
q2 = q.copy(); q2[:23] += rng.uniform(-0.2, 0.4, 23); q2[25] -= 0.003
vv, aa, cc = rng.normal(0, 0.5, 35), rng.uniform(0, 1, 39), rng.uniform(0, 1, 39)
fwd_compare("hand rand f64", mj, q=q2, v=vv, act=aa, ctrl=cc)
fwd_compare("hand rand f32", mj, q=q2, v=vv, act=aa, ctrl=cc, dtype=native.MYO_F32)


def traj(name, mj, nsteps, q=None, dtype=native.MYO_F64, integrator=None):
    cm = compile_model(mj, integrator=integrator, unsupported_contacts="drop")
    om = OracleModel(cm.to_blob()); d = OracleData(om)
    nm = native.Model(cm, lib); b = native.Batch(nm, None, 2, 0, 0, dtype)
    nq, nv, na, nu = om.nq, om.nv, om.na, om.nu
    if q is not None: d.qpos[:] = q
    b.set_state(T(np.tile(d.qpos, (2, 1))), T(np.zeros((2, nv))), T(np.zeros((2, na))), T(np.zeros(2)))
    qp = torch.zeros((2, nq), dtype=torch.float64, device=dev); qv = torch.zeros((2, nv), dtype=torch.float64, device=dev)
    for i in range(nsteps):
        if i % 20 == 0: c = rng.uniform(0, 1, (1, nu))
        d.ctrl[:] = c[0]; d.step(); b.physics_step(T(np.tile(c, (2, 1))), 1)
    b.get_state(qp, qv, None, None); torch.cuda.synchronize()
    print(name, nsteps, "dq", np.abs(qp[1].cpu().numpy() - d.qpos).max(), "dv", np.abs(qv[1].cpu().numpy() - d.qvel).max(), "nefc", d.nefc)


traj("finger rk4", mjf, 200, integrator=1)
traj("hand euler f64", mj, 100, q=q)
traj("hand rk4 f64", mj, 40, q=q, integrator=1)
traj("hand euler f32", mj, 20, q=q, dtype=native.MYO_F32)

# task step parity + timing
cm = compile_model(mj)
for dtype, dn in ((native.MYO_F64, "f64"), (native.MYO_F32, "f32")):
    om = OracleModel(cm.to_blob()); d = OracleData(om)
    cfgc = make_task_cfg("CustomMyoBaodingBallsP1", cm)
    nm = native.Model(cm, lib); N = 8
    b = native.Batch(nm, cfgc, N, 0, 123, dtype)
    obs = torch.zeros((N, 86), dtype=torch.float32, device=dev); b.reset(None, obs); torch.cuda.synchronize()
    g = np.load("tests/golden/reset_obs_golden.npy")
    print(dn, "reset obs vs golden", np.abs(obs[0].cpu().numpy() - g).max())
    ocfg = make_cfg(task_ids(cm)); d.reset(); d.qpos[0] = -1.57
    st = BaodingState(); st.which_task = 2; st.counter = 0; st.start_angle[0] = 3 * np.pi / 4; st.start_angle[1] = -np.pi / 4
    st.x_radius = 0.025; st.y_radius = 0.028; st.time_period = 5
    rew = torch.zeros(N, dtype=torch.float32, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
    trunc = torch.zeros(N, dtype=torch.uint8, device=dev); tobs = torch.zeros((N, 86), dtype=torch.float32, device=dev)
    comps = torch.zeros((N, 8), dtype=torch.float32, device=dev); ep = torch.zeros((N, 2), dtype=torch.float32, device=dev)
    for i in range(30):
        a = np.clip(rng.normal(0, 0.3, (1, 39)), -1, 1).astype(np.float32)
        b.step(T(np.tile(a, (N, 1)), torch.float32), obs, rew, done, trunc, tobs, comps, ep); torch.cuda.synchronize()
        oo, cc_ = baoding_step(d, ocfg, st, a[0])
        if i % 6 == 0: print(dn, i, "obs err", np.abs(obs[5].cpu().numpy() - oo).max(), "rew", float(rew[5]), cc_[7], "done", int(done[5]))

for integ, iname in ((0, "euler"), (1, "rk4")):
    cmi = compile_model(mj, integrator=integ)
    for dtype, dn in ((native.MYO_F32, "f32"), (native.MYO_F64, "f64")):
        for N in (4096, 8192):
            nm = native.Model(cmi, lib)
            cfgc = make_task_cfg("CustomMyoBaodingBallsP1", cmi)
            b = native.Batch(nm, cfgc, N, 0, 1, dtype)
            obs = torch.zeros((N, 86), dtype=torch.float32, device=dev)
            rew = torch.zeros(N, dtype=torch.float32, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
            b.reset(None, obs)
            act = torch.clamp(torch.randn((N, 39), device=dev) * 0.135, -1, 1)
            for _ in range(3): b.step(act, obs, rew, done)
            torch.cuda.synchronize(); t0 = time.time(); K = 10
            ndone = 0
            for _ in range(K):
                act = torch.clamp(torch.randn((N, 39), device=dev) * 0.135, -1, 1)
                b.step(act, obs, rew, done); ndone += int(done.sum())
            torch.cuda.synchronize(); dt = (time.time() - t0) / K
            print(f"TIMING {iname} {dn} N={N}: {dt*1e3:.2f} ms/step -> {N/dt:,.0f} env-steps/s  lds={b.lds_bytes} dones/step={ndone/K:.1f} finite={bool(torch.isfinite(obs).all())}")
            b.close()
