#!/usr/bin/env python3
"""Dump a per-substep trajectory of a MuJoCo model for a seeded control stream, in the golden format the
parity tests read (tests/test_true_reference.py picks up every tests/golden/mujoco_traj_*.npz).

WHY: the reference's physics (MuJoCo 2.1 through free-mujoco-py==2.1.6, /root/reference/requirements.txt:41)
is absent from the reference tree and from the build image, so this repo's CPU oracle is a restatement whose
STEPPING is unpinned (oracle/myo_oracle.h).  A user who has MuJoCo can produce true trajectory goldens with
this script; committing the resulting .npz pins the oracle (and through it the HIP stepper).

    python tools/dump_mujoco_trajectory.py --mjb path/to/myo_hand_baoding.mjb --out tests/golden/mujoco_traj_hand.npz \
        [--backend mujoco_py|mujoco|oracle] [--substeps 2000] [--seed 0] [--hold 10] [--init-qpos0 -1.57]

Backends
  mujoco_py  free-mujoco-py / mujoco-py 2.1 (the reference's own binding): load_model_from_mjb, MjSim.step
  mujoco     DeepMind's `mujoco` python package (>= 2.1.2 reads .mjb through MjModel.from_binary_path); its
             solver defaults equal MuJoCo 2.1's for the options MyoSuite models set
  oracle     this repo's CPU restatement — a stand-in producer used by the unit test of the file format
             (a file written with it pins nothing and says so in its metadata)

File format (numpy .npz; T = number of substeps)
  meta          json string: producer, producer_version, model (file name), nq, nv, nu, na, timestep, integrator,
                seed, hold, is_true_reference
  ctrl          float64 [T, nu]      control applied during substep t (uniform(0,1), held for `hold` substeps)
  qpos, qvel    float64 [T+1, nq|nv] state BEFORE substep t (row 0 = initial state, row T = final state)
  act           float64 [T+1, na]
  qacc_warmstart float64 [T+1, nv]   (mjData.qacc_warmstart before substep t: part of the integration state)
  ncon, nefc, solver_iter  int32 [T] after the forward pass of substep t
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FORMAT_KEYS = ("meta", "ctrl", "qpos", "qvel", "act", "qacc_warmstart", "ncon", "nefc", "solver_iter")


def control_stream(nu, substeps, seed, hold):
    rng = np.random.RandomState(seed)
    ctrl = np.zeros((substeps, nu))
    for t in range(substeps):
        if t % hold == 0:
            c = rng.uniform(0, 1, nu)
        ctrl[t] = c
    return ctrl


class _MujocoPy:
    name = "mujoco_py"

    def __init__(self, path):
        import mujoco_py
        self.version = getattr(mujoco_py, "__version__", "?")
        self.model = mujoco_py.load_model_from_mjb(path)
        self.sim = mujoco_py.MjSim(self.model)
        m = self.model
        self.sizes = dict(nq=m.nq, nv=m.nv, nu=m.nu, na=m.na, timestep=float(m.opt.timestep), integrator=int(m.opt.integrator))

    def set_qpos(self, q):
        self.sim.data.qpos[:] = q
        self.sim.forward()

    def state(self):
        d = self.sim.data
        return d.qpos.copy(), d.qvel.copy(), (d.act.copy() if self.model.na else np.zeros(0)), d.qacc_warmstart.copy()

    def step(self, ctrl):
        d = self.sim.data
        d.ctrl[:] = ctrl
        self.sim.step()
        return int(d.ncon), int(d.nefc), int(d.solver_iter)

    def qpos0(self):
        return self.model.qpos0.copy()


class _Mujoco:
    name = "mujoco"

    def __init__(self, path):
        import mujoco
        self.mj = mujoco
        self.version = mujoco.__version__
        self.model = mujoco.MjModel.from_binary_path(path)
        self.data = mujoco.MjData(self.model)
        m = self.model
        self.sizes = dict(nq=m.nq, nv=m.nv, nu=m.nu, na=m.na, timestep=float(m.opt.timestep), integrator=int(m.opt.integrator))

    def set_qpos(self, q):
        self.data.qpos[:] = q
        self.mj.mj_forward(self.model, self.data)

    def state(self):
        d = self.data
        return d.qpos.copy(), d.qvel.copy(), d.act.copy(), d.qacc_warmstart.copy()

    def step(self, ctrl):
        d = self.data
        d.ctrl[:] = ctrl
        self.mj.mj_step(self.model, d)
        it = d.solver_iter
        return int(d.ncon), int(d.nefc), int(it[0] if hasattr(it, "__len__") else it)

    def qpos0(self):
        return self.model.qpos0.copy()


class _Oracle:
    """Stand-in producer (NOT a true reference): the repo's own CPU restatement."""
    name = "oracle"

    def __init__(self, path):
        from myochallenge_amd.mjb import load_mjb
        from myochallenge_amd.model import compile_model
        from myochallenge_amd.synth_hand import synthetic_hand
        from oracle.oracle import OracleData, OracleModel
        self.version = "myochallenge_amd oracle (unpinned stepping)"
        mj = synthetic_hand() if path == "synthetic_hand" else load_mjb(path)
        self.mjb = mj
        cm = compile_model(mj, unsupported_contacts="drop")
        self.om = OracleModel(cm.to_blob())
        self.d = OracleData(self.om)
        self.sizes = dict(nq=self.om.nq, nv=self.om.nv, nu=self.om.nu, na=self.om.na, timestep=float(mj.opt["timestep"]),
                          integrator=int(mj.opt["integrator"]))

    def set_qpos(self, q):
        self.d.qpos[:] = q

    def state(self):
        d = self.d
        return np.array(d.qpos), np.array(d.qvel), np.array(d.act), np.array(d.qacc_warmstart)

    def step(self, ctrl):
        self.d.ctrl[:] = ctrl
        self.d.step()
        return self.d.ncon, self.d.nefc, self.d.solver_iter

    def qpos0(self):
        return np.array(self.mjb.qpos0)


BACKENDS = {"mujoco_py": _MujocoPy, "mujoco": _Mujoco, "oracle": _Oracle}


def dump(backend, path, out, substeps=2000, seed=0, hold=10, init_qpos0=None, n_hand=23):
    be = BACKENDS[backend](path)
    sz = be.sizes
    q = be.qpos0()
    if init_qpos0 is not None:      # the Baoding reset pose: init_qpos[:-14] = 0, init_qpos[0] = -1.57 (baoding.py:281-283)
        q[:n_hand] = 0
        q[0] = init_qpos0
    be.set_qpos(q)
    ctrl = control_stream(sz["nu"], substeps, seed, hold)
    rows = {k: [] for k in ("qpos", "qvel", "act", "qacc_warmstart")}
    cnt = {k: [] for k in ("ncon", "nefc", "solver_iter")}
    for t in range(substeps + 1):
        s = be.state()
        for k, v in zip(rows, s):
            rows[k].append(v)
        if t == substeps:
            break
        c = be.step(ctrl[t])
        for k, v in zip(cnt, c):
            cnt[k].append(v)
    meta = dict(producer=be.name, producer_version=str(be.version), model=os.path.basename(path), seed=seed, hold=hold,
                is_true_reference=be.name != "oracle", **sz)
    np.savez_compressed(out, meta=json.dumps(meta), ctrl=ctrl, **{k: np.array(v, np.float64) for k, v in rows.items()},
                        **{k: np.array(v, np.int32) for k, v in cnt.items()})
    return meta


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mjb", required=True, help="model file (.mjb); 'synthetic_hand' with --backend oracle")
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="mujoco_py", choices=sorted(BACKENDS))
    ap.add_argument("--substeps", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--hold", type=int, default=10, help="substeps a control vector is held (frame_skip)")
    ap.add_argument("--init-qpos0", type=float, default=None, help="Baoding reset pose: hand joints 0, qpos[0] = this (-1.57)")
    a = ap.parse_args()
    meta = dump(a.backend, a.mjb, a.out, a.substeps, a.seed, a.hold, a.init_qpos0)
    print("wrote", a.out, meta)


if __name__ == "__main__":
    main()
