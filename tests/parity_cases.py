"""Parity cases shared by tests/test_emu_parity.py (CPU: lane-serial emulation of the kernel
source) and tests/test_gpu_parity.py (-m gpu: the HIP kernels on an MI355X).  Every case goes
through the C ABI and compares with the oracle on the same seeded inputs.

Tolerances: fp64 stepper vs fp64 oracle 1e-9 relative (different summation order only);
mixed stepper (MYO_MIXED: fp64 state / kinematic chain / contact distances, fp32 dynamics) 1e-4 relative
over whole episodes (north_star's stated tolerance), see case_episode_trajectory.
"""
import numpy as np

from helpers import Mem, default_state, forward_dump, oracle_for, rel_err
from oracle.oracle import OracleData
from helpers import Mem as _Mem  # noqa: F401
from myochallenge_amd import native
from myochallenge_amd.envs.config import make_task_cfg, task_ids
from oracle.oracle import baoding_step, make_cfg

STAGES = ("ten_length", "ten_J", "M", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "qacc",
          "actuator_force", "act_dot")


def hand_states(mj, n, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        q = mj.qpos0.copy(); q[0] = -1.57
        if i:
            q[:23] += rng.uniform(-0.2, 0.4, 23)
            q[25] -= rng.uniform(0, 0.004)
        out.append((q, rng.normal(0, 0.5 * (i > 0), 35), rng.uniform(0, 1, 39) * (i > 0), rng.uniform(0, 1, 39)))
    return out


def case_forward_stages(lib, models, dtype, tol):
    mem = Mem(lib)
    rng = np.random.RandomState(3)
    cases = [("finger", models["finger"], np.array([0.5, 1.2, 1.2, 1.1]), rng.normal(0, 1, 4), rng.uniform(0, 1, 5), rng.uniform(0, 1, 5)),
             ("finger", models["finger"], rng.uniform(-0.3, 0.9, 4), rng.normal(0, 1, 4), rng.uniform(0, 1, 5), rng.uniform(0, 1, 5)),
             ("load", models["load"], np.array([-0.006]), np.array([0.1]), np.array([0.3]), np.array([0.8]))]
    cases += [("hand", models["hand"], *s) for s in hand_states(models["hand"], 4)]
    for name, mj, q, v, a, c in cases:
        cm, om, d = oracle_for(mj)
        d.qpos[:], d.qvel[:], d.act[:], d.ctrl[:] = q, v, a, c
        get, b = forward_dump(lib, mem, cm, q, v, a, c, dtype)
        d.forward()
        cnt = get("counts", 4)
        same = (int(cnt[0]), int(cnt[1]), int(cnt[3])) == (d.ncon, d.nefc, d.nl)
        assert same, (name, cnt)                    # contact / limit activation is decided in fp64 by both steppers
        for st in STAGES:
            ref = np.array(getattr(d, st))
            assert rel_err(get(st, ref.size), ref) < tol, (name, st, rel_err(get(st, ref.size), ref))
        b.close()


def case_trajectory(lib, mj, nsteps, dtype, tol, integrator=None, q0=None, seed=0):
    mem = Mem(lib)
    cm, om, d = oracle_for(mj, integrator=integrator)
    n = 3
    b = native.Batch(native.Model(cm, lib), None, n, 0, 0, dtype)
    if q0 is not None:
        d.qpos[:] = q0
    b.set_state(mem.arr(np.tile(d.qpos, (n, 1))), mem.zeros((n, om.nv)), mem.zeros((n, om.na)), mem.zeros(n))
    rng = np.random.RandomState(seed)
    qp, qv, ac, tt = mem.zeros((n, om.nq)), mem.zeros((n, om.nv)), mem.zeros((n, om.na)), mem.zeros(n)
    for i in range(nsteps):
        if i % 10 == 0:
            c = rng.uniform(0, 1, om.nu)
        d.ctrl[:] = c
        d.step()
        b.physics_step(mem.arr(np.tile(c, (n, 1))), 1)
    b.get_state(qp, qv, ac, tt)
    qp, qv, ac, tt = (mem.host(x) for x in (qp, qv, ac, tt))
    assert np.array_equal(qp[0], qp[2]) and np.array_equal(qv[0], qv[2])      # envs are independent + deterministic
    assert rel_err(qp[1], d.qpos) < tol and np.abs(qv[1] - d.qvel).max() < tol * max(1.0, np.abs(d.qvel).max())
    assert rel_err(ac[1], d.act) < tol and abs(tt[1] - d.arr("time")[0]) < 1e-9
    b.close()


def episode_drift(lib, mj, dtype, streams, nsteps=200, integrator=None, env_name="CustomMyoBaodingBallsP1", ball_d=None, resync=False, seed=77, **env_kw):
    """The trajectory-parity measurement the contract names (BASELINE.json north_star: "state-trajectory match
    to the reference CPU step on identical seeds"): one env per action stream, `nsteps` env steps
    (x frame_skip substeps) with the VecEnv's auto-reset, HIP (or emulation) stepper against the oracle stepped
    with the same float32 actions.  streams = [(sigma, seed), ...]: actions ~ clip(N(0, sigma), -1, 1).
    ball_d ([n, 10], optional): per-env ball mass / friction / size injected into both sides (P2 physics).

    Returns per stream: err_q[t] = max |qpos - qpos_oracle| / max |qpos_oracle|  (the balls' z ~ 1.45 sets the
    scale), err_obs[t] = max |obs - obs_oracle| on the 86-vector (terminal observation on the steps that end an
    episode), ends = steps at which an episode ended.  If the two sides end an episode on different steps the
    stream's errors are 1.0 from that step on.

    resync=True (envs whose reset DRAWS: CustomBaodingP2Env's goal / ball mass / friction / size randomisation,
    /root/reference/src/envs/baoding.py:494-647): after the first reset and after every auto-reset the oracle twin is
    rebuilt from the device's post-reset state and the episode's draws (task scalars, ball parameters) read back
    through the C ABI, so that both sides step the SAME episode; seed-for-seed RNG parity is out of scope."""
    mem = Mem(lib)
    cm, om, _ = oracle_for(mj, integrator=integrator)
    n = len(streams)
    tc = make_task_cfg(env_name, cm, **env_kw)
    b = native.Batch(native.Model(cm, lib), tc, n, 0, seed, dtype)
    obs = mem.zeros((n, 86), np.float32)
    b.reset(None, obs)
    if ball_d is not None:
        b.set_task(None, None, mem.arr(ball_d))
    ocfg = make_cfg(task_ids(cm), drop_th=tc.drop_th, proximity_th=tc.proximity_th,
                    weights={k: tc.weights[i] for i, k in enumerate(("pos_dist_1", "pos_dist_2", "act_reg", "alive", "sparse", "solved", "done"))})
    s_qp, s_qv, s_ac, s_tm = mem.zeros((n, om.nq)), mem.zeros((n, om.nv)), mem.zeros((n, om.na)), mem.zeros(n)
    s_ti, s_td, s_bd, s_w = mem.zeros((n, 2), np.int32), mem.zeros((n, 9)), mem.zeros((n, 10)), mem.zeros((n, om.nv))

    def twin_from_device(e):
        """oracle twin of env e: the device's (post-reset) state and the episode's draws"""
        b.get_state(s_qp, s_qv, s_ac, s_tm); b.get_task(s_ti, s_td, s_bd); b.warmstart(s_w, None)
        h = [mem.host(x) for x in (s_qp, s_qv, s_ac, s_tm, s_ti, s_td, s_bd, s_w)]
        d = OracleData(om)
        d.reset()
        d.qpos[:], d.qvel[:], d.act[:] = h[0][e], h[1][e], h[2][e]
        d.arr("time")[0] = h[3][e]
        d.arr("qacc_warmstart")[:] = h[7][e]
        d.set_ball_params(ocfg, h[6][e])
        sp = d.arr("site_pos")                           # Task.HOLD never moves the target sites: they keep the last episode's xy
        sp[3 * ocfg.target1_sid:3 * ocfg.target1_sid + 2] = h[5][e, 5:7]
        sp[3 * ocfg.target2_sid:3 * ocfg.target2_sid + 2] = h[5][e, 7:9]
        st = default_state(which=int(h[4][e, 0]), period=h[5][e, 4], xr=h[5][e, 2], yr=h[5][e, 3], s1=h[5][e, 0], s2=h[5][e, 1])
        st.counter = int(h[4][e, 1])
        return [d, st, 0]

    orc = []
    for e in range(n):
        if resync:
            orc.append(twin_from_device(e))
            continue
        d = OracleData(om)
        d.reset(); d.qpos[:23] = 0; d.qpos[0] = -1.57
        if ball_d is not None:
            d.set_ball_params(ocfg, ball_d[e])
        orc.append([d, default_state(), 0])
    rngs = [np.random.RandomState(seed) for _, seed in streams]
    rew, done, trunc = mem.zeros(n, np.float32), mem.zeros(n, np.uint8), mem.zeros(n, np.uint8)
    term = mem.zeros((n, 86), np.float32)
    qp = mem.zeros((n, om.nq))
    err_q, err_obs, ends = np.zeros((n, nsteps)), np.zeros((n, nsteps)), [[] for _ in range(n)]
    split = [None] * n
    for t in range(nsteps):
        a = np.stack([np.clip(r.normal(0, sg, 39), -1, 1) for r, (sg, _) in zip(rngs, streams)]).astype(np.float32)
        b.step(mem.arr(a, np.float32), obs, rew, done, trunc, term)
        b.get_state(qp)
        h_obs, h_term, h_done, h_trunc, h_qp = (mem.host(x) for x in (obs, term, done, trunc, qp))
        for e in range(n):
            if split[e] is not None:                     # the two sides ended an episode on different steps:
                err_q[e, t] = err_obs[e, t] = 1.0        # nothing left to compare on this stream
                continue
            d, st, el = orc[e]
            o, c = baoding_step(d, ocfg, st, a[e])
            el += 1
            o_done = bool(c[6]) or el >= tc.max_episode_steps
            if bool(h_done[e]) != o_done:
                split[e] = t
                err_q[e, t] = err_obs[e, t] = 1.0
                continue
            if o_done:
                assert bool(h_trunc[e]) == (not c[6])
                err_obs[e, t] = np.abs(h_term[e] - o).max()
                err_q[e, t] = err_q[e, t - 1] if t else 0.0      # the state was reset on the device: carry the last value
                ends[e].append(t)
                if resync:
                    orc[e] = twin_from_device(e)
                    continue
                d.reset(); d.qpos[:23] = 0; d.qpos[0] = -1.57    # P1 reset without noise / RSI is deterministic
                if ball_d is not None:
                    d.set_ball_params(ocfg, ball_d[e])
                orc[e] = [d, default_state(), 0]
            else:
                err_obs[e, t] = np.abs(h_obs[e] - o).max()
                err_q[e, t] = np.abs(h_qp[e] - d.qpos).max() / np.abs(d.qpos).max()
                orc[e][2] = el
    b.close()
    return {"streams": [list(x) for x in streams], "err_qpos_rel": err_q, "err_obs_abs": err_obs, "episode_ends": ends,
            "episode_end_disagreement_at": split}


def local_error(lib, mj, dtype, streams, nsteps=200, integrator=None, seed=77):
    """LOCAL (one-env-step) error of a stepper along the oracle's trajectory (VERDICT r03 "weak" 1: the whole-episode
    drift of the mixed stepper mixes its own error with the trajectory's sensitivity; this separates them).  The oracle
    walks the 16 seeded action streams of episode_drift (CustomMyoBaodingBallsP1, auto-resets included); BEFORE every env
    step the device is put on the oracle's state (qpos, qvel, act, time, qacc_warmstart — the task counters advance in
    lockstep), both take the step, and the resulting qpos / qvel / obs are compared.  Nothing is carried from one step to the
    next on the device, so err[t] is what ONE env step (10 substeps) of the stepper adds.
    Returns err_qpos_rel [n, nsteps], err_qvel_abs, err_obs_abs, done_disagreements (list of (stream, step))."""
    mem = Mem(lib)
    cm, om, _ = oracle_for(mj, integrator=integrator)
    n = len(streams)
    tc = make_task_cfg("CustomMyoBaodingBallsP1", cm)
    b = native.Batch(native.Model(cm, lib), tc, n, 0, seed, dtype)
    obs = mem.zeros((n, 86), np.float32)
    b.reset(None, obs)
    ocfg = make_cfg(task_ids(cm), drop_th=tc.drop_th, proximity_th=tc.proximity_th,
                    weights={k: tc.weights[i] for i, k in enumerate(("pos_dist_1", "pos_dist_2", "act_reg", "alive", "sparse", "solved", "done"))})
    orc = []
    for e in range(n):
        d = OracleData(om)
        d.reset(); d.qpos[:23] = 0; d.qpos[0] = -1.57
        orc.append([d, default_state(), 0])
    rngs = [np.random.RandomState(sd) for _, sd in streams]
    rew, done, trunc = mem.zeros(n, np.float32), mem.zeros(n, np.uint8), mem.zeros(n, np.uint8)
    term = mem.zeros((n, 86), np.float32)
    qp, qv = mem.zeros((n, om.nq)), mem.zeros((n, om.nv))
    err_q, err_v, err_o = np.zeros((n, nsteps)), np.zeros((n, nsteps)), np.zeros((n, nsteps))
    disagree, n_ends = [], 0
    for t in range(nsteps):
        # the device starts this step from the oracle's state
        b.set_state(mem.arr(np.stack([o[0].qpos for o in orc])), mem.arr(np.stack([o[0].qvel for o in orc])),
                    mem.arr(np.stack([o[0].act for o in orc])), mem.arr(np.array([o[0].arr("time")[0] for o in orc])))
        b.warmstart(None, mem.arr(np.stack([np.array(o[0].arr("qacc_warmstart")) for o in orc])))
        a = np.stack([np.clip(r.normal(0, sg, 39), -1, 1) for r, (sg, _) in zip(rngs, streams)]).astype(np.float32)
        b.step(mem.arr(a, np.float32), obs, rew, done, trunc, term)
        b.get_state(qp, qv)
        h_obs, h_term, h_done, h_qp, h_qv = (mem.host(x) for x in (obs, term, done, qp, qv))
        for e in range(n):
            d, st, el = orc[e]
            o, c = baoding_step(d, ocfg, st, a[e])
            el += 1
            o_done = bool(c[6]) or el >= tc.max_episode_steps
            if bool(h_done[e]) != o_done:
                disagree.append((e, t))
                err_q[e, t] = err_v[e, t] = err_o[e, t] = 1.0
            elif o_done:
                err_o[e, t] = np.abs(h_term[e] - o).max()       # (the device state has been reset: the terminal observation is what is left of the step)
            else:
                err_o[e, t] = np.abs(h_obs[e] - o).max()
                err_q[e, t] = np.abs(h_qp[e] - d.qpos).max() / np.abs(d.qpos).max()
                err_v[e, t] = np.abs(h_qv[e] - d.qvel).max()
            if o_done or bool(h_done[e]):
                n_ends += 1
                # both sides start the next episode from the deterministic P1 reset; a device that did NOT reset is put
                # there too (set_state above), its task counters by an explicit reset of that env
                if not bool(h_done[e]):
                    m = np.zeros(n, np.uint8); m[e] = 1
                    b.reset(mem.arr(m, np.uint8), obs)
                d.reset(); d.qpos[:23] = 0; d.qpos[0] = -1.57
                orc[e] = [d, default_state(), 0]
            else:
                orc[e][2] = el
    b.close()
    return {"streams": [list(x) for x in streams], "err_qpos_rel": err_q, "err_qvel_abs": err_v, "err_obs_abs": err_o,
            "done_disagreements": disagree, "episode_ends": n_ends}


def write_drift_record(r, path, dtype_name, integrator_name, nsteps):
    """Drift table of episode_drift as JSON (every 10th step) — test evidence, copied into profiles/ per round."""
    import json
    import os
    os.makedirs(os.path.dirname(path), exist_ok=True)
    f3 = lambda v: float("%.3g" % v)
    json.dump({"what": "HIP stepper vs oracle, tests/parity_cases.episode_drift: err_qpos_rel = max|qpos - qpos_oracle| / max|qpos_oracle|, "
                       "err_obs_abs = max|obs - obs_oracle| (float32 observation, terminal observation on steps that end an episode)",
               "dtype": dtype_name, "integrator": integrator_name, "env_steps": nsteps, "substeps_per_env_step": 10,
               "streams (action sigma, seed)": r["streams"], "episode_ends": r["episode_ends"],
               "episode_end_disagreement_at": r["episode_end_disagreement_at"],
               "max_err_qpos_rel": [f3(v) for v in r["err_qpos_rel"].max(1)], "max_err_obs_abs": [f3(v) for v in r["err_obs_abs"].max(1)],
               "steps_above_1e-4 (step, err_qpos_rel)": [[(int(t), f3(row[t])) for t in np.nonzero(row > 1e-4)[0][:12]] for row in r["err_qpos_rel"]],
               "err_qpos_rel_every_10th_step": [[f3(v) for v in row[9::10]] for row in r["err_qpos_rel"]],
               "err_obs_abs_every_10th_step": [[f3(v) for v in row[9::10]] for row in r["err_obs_abs"]]}, open(path, "w"), indent=1)


def case_episode_trajectory(lib, mj, dtype, streams, tol, nsteps=200, integrator=None):
    """assert rel_err(qpos) <= tol and obs error <= max(tol, 1e-7: float32 obs) at EVERY step of `nsteps` env steps, for every stream."""
    r = episode_drift(lib, mj, dtype, streams, nsteps, integrator)
    worst_q, worst_o = r["err_qpos_rel"].max(), r["err_obs_abs"].max()
    assert worst_q <= tol and worst_o <= max(tol, 1e-7), (worst_q, worst_o, r["err_qpos_rel"].max(1), r["err_obs_abs"].max(1))
    return r


def case_task_step(lib, models, dtype, tol, nsteps=25):
    """env.step parity: obs (86), reward components, done — against the oracle's Baoding step."""
    mem = Mem(lib)
    cm, om, d = oracle_for(models["hand"])
    weights = {"pos_dist_1": 1, "pos_dist_2": 1, "act_reg": 0.5, "alive": 1, "solved": 5, "done": -2, "sparse": 0.1}
    tc = make_task_cfg("CustomMyoBaodingBallsP1", cm, weighted_reward_keys=weights, drop_th=1.3)
    n = 4
    b = native.Batch(native.Model(cm, lib), tc, n, 0, 11, dtype)
    obs = mem.zeros((n, 86), np.float32)
    b.reset(None, obs)
    ocfg = make_cfg(task_ids(cm), drop_th=1.3, weights=weights)
    d.reset(); d.qpos[0] = -1.57
    st = default_state()
    rew, done, trunc = mem.zeros(n, np.float32), mem.zeros(n, np.uint8), mem.zeros(n, np.uint8)
    term, comps, ep = mem.zeros((n, 86), np.float32), mem.zeros((n, 8), np.float32), mem.zeros((n, 2), np.float32)
    rng = np.random.RandomState(5)
    ret = 0.0
    for i in range(nsteps):
        a = np.clip(rng.normal(0, 0.4, 39), -1.2, 1.2).astype(np.float32)
        b.step(mem.arr(np.tile(a, (n, 1)), np.float32), obs, rew, done, trunc, term, comps, ep)
        o, c = baoding_step(d, ocfg, st, a)
        ret += c[7]
        dn = mem.host(done)
        got = mem.host(term)[2] if dn[2] else mem.host(obs)[2]
        assert np.abs(got - o).max() < max(tol, 2e-7), (i, np.abs(got - o).max())      # float32 outputs: 1 ulp of 1.45 is 1.2e-7
        assert np.abs(mem.host(comps)[2] - c).max() < max(tol, 2e-6) * max(1.0, np.abs(c).max()), (i, mem.host(comps)[2], c)
        assert bool(dn[2]) == bool(c[6])
        if dn[2]:
            e = mem.host(ep)[2]
            assert abs(e[0] - ret) < 1e-3 * max(1, abs(ret)) and int(e[1]) == i + 1
            break
    b.close()


def case_vecenv_protocol(lib, models, dtype):
    """TimeLimit.truncated at 200 steps, terminal_observation, auto-reset, Monitor episode stats,
    fall termination via drop_th — SubprocVecEnv semantics."""
    mem = Mem(lib)
    import copy
    mj = copy.deepcopy(models["hand"])
    mj.arrays["geom_contype"][:] = 0; mj.arrays["geom_conaffinity"][:] = 0      # balls fall freely
    from myochallenge_amd.model import compile_model
    cm = compile_model(mj)
    n = 2
    tc = make_task_cfg("CustomMyoBaodingBallsP1", cm)
    b = native.Batch(native.Model(cm, lib), tc, n, 0, 3, dtype)
    obs = mem.zeros((n, 86), np.float32); b.reset(None, obs)
    first = mem.host(obs).copy()
    rew, done, trunc = mem.zeros(n, np.float32), mem.zeros(n, np.uint8), mem.zeros(n, np.uint8)
    term, comps, ep = mem.zeros((n, 86), np.float32), mem.zeros((n, 8), np.float32), mem.zeros((n, 2), np.float32)
    act = mem.zeros((n, 39), np.float32)
    # free fall from z=1.452: z < drop_th(1.25) after sqrt(2*0.2/9.81)=0.202 s = 11 env steps
    for i in range(1, 30):
        b.step(act, obs, rew, done, trunc, term, comps, ep)
        if mem.host(done)[0]:
            break
    assert 9 <= i <= 12
    assert not mem.host(trunc)[0] and mem.host(comps)[0, 6] == 1.0          # done by fall, not truncated
    assert mem.host(term)[0, 25] < 1.25 or mem.host(term)[0, 31] < 1.25      # terminal_observation is pre-reset
    o_now = mem.host(obs)[0]
    keep = np.r_[0:35, 47:86]            # target sites keep their last position until the next step
    assert np.abs(o_now[keep] - first[0][keep]).max() < 1e-6                 # returned obs is the reset obs
    assert np.abs(o_now[35:41] - first[0][35:41]).max() > 1e-4               # (reference quirk)
    assert int(mem.host(ep)[0, 1]) == i
    b.close()
    # truncation: a hand that holds still with balls resting (no gravity) for 200 steps
    mj2 = copy.deepcopy(models["hand"])
    mj2.opt = dict(mj2.opt); mj2.opt["gravity"] = [0.0, 0.0, 0.0]
    cm2 = compile_model(mj2)
    b = native.Batch(native.Model(cm2, lib), make_task_cfg("CustomMyoBaodingBallsP1", cm2), 1, 0, 3, dtype)
    obs = mem.zeros((1, 86), np.float32); b.reset(None, obs)
    rew, done, trunc = mem.zeros(1, np.float32), mem.zeros(1, np.uint8), mem.zeros(1, np.uint8)
    term, comps, ep = mem.zeros((1, 86), np.float32), mem.zeros((1, 8), np.float32), mem.zeros((1, 2), np.float32)
    act = mem.arr(np.full((1, 39), -1.0), np.float32)
    for i in range(1, 201):
        b.step(act, obs, rew, done, trunc, term, comps, ep)
        if mem.host(done)[0]:
            break
    assert i == 200 and mem.host(trunc)[0] == 1 and mem.host(comps)[0, 6] == 0.0
    assert int(mem.host(ep)[0, 1]) == 200
    b.step(act, obs, rew, done, trunc, term, comps, ep)
    assert mem.host(done)[0] == 0
    b.close()


def case_reset_logic(lib, models, dtype):
    """Device reset vs the reference's reset logic (structure pinned by reset_logic_goldens.json):
    which qpos slots change, parameter ranges, RSI teleport, determinism per seed."""
    mem = Mem(lib)
    from myochallenge_amd.model import compile_model
    cm = compile_model(models["hand"])
    n = 64
    init = models["hand"].qpos0.copy(); init[:23] = 0; init[0] = -1.57

    def run(name, seed=5, **kw):
        b = native.Batch(native.Model(cm, lib), make_task_cfg(name, cm, **kw), n, 0, seed, dtype)
        obs = mem.zeros((n, 86), np.float32); b.reset(None, obs)
        qp, qv, ac, tt = mem.zeros((n, 37)), mem.zeros((n, 35)), mem.zeros((n, 39)), mem.zeros(n)
        b.get_state(qp, qv, ac, tt)
        ti, td, bd = mem.zeros((n, 2), np.int32), mem.zeros((n, 9)), mem.zeros((n, 10))
        b.get_task(ti, td, bd)
        out = tuple(mem.host(x).copy() for x in (obs, qp, qv, ac, tt, ti, td, bd))
        b.close()
        return out

    # P1 defaults: nothing random but the (degenerate) ranges; state = init
    obs, qp, qv, ac, tt, ti, td, bd = run("CustomMyoBaodingBallsP1")
    assert np.allclose(qp, init) and not qv.any() and not ac.any() and not tt.any()
    assert (ti[:, 0] == 2).all() and (ti[:, 1] == 0).all()                       # CCW, counter 0
    assert np.allclose(td[:, :5], [3 * np.pi / 4, -np.pi / 4, 0.025, 0.028, 5.0])
    # P1 noise: exactly the reference's slots (baoding.py:96-144,192-197)
    obs, qp, qv, ac, tt, ti, td, bd = run("CustomMyoBaodingBallsP1", noise_palm=1, noise_fingers=1, noise_balls=0.01, task="random")
    changed = np.abs(qp - init).max(0) > 0
    expect = np.zeros(37, bool); expect[:23] = True; expect[[23, 24, 25, 30, 31, 32]] = True
    assert (changed == expect).all()
    assert (qp[:, 0] >= -np.pi / 2).all() and (qp[:, 0] <= -np.pi / 2 + np.pi / 18).all()
    assert np.abs(qp[:, 1:3]).max() <= np.pi / 18 and np.abs(qp[:, 3:7]).max() <= np.pi / 18
    assert np.ptp(qp[:, 3:7], axis=1).max() == 0                                 # ONE draw broadcast to the thumb
    fl = [7, 9, 10, 11, 13, 14, 15, 17, 18, 19, 21, 22]
    assert (qp[:, fl] >= 0).all() and (qp[:, fl] <= np.pi / 6).all() and np.ptp(qp[:, fl], axis=1).max() == 0
    assert np.abs(qp[:, [8, 12, 16, 20]]).max() <= np.pi / 36
    assert set(np.unique(ti[:, 0])) == {0, 1, 2}                                 # random.choice(list(Task))
    # P2 registration defaults: physics randomisation ranges (baoding.py:559-604)
    obs, qp, qv, ac, tt, ti, td, bd = run("CustomMyoBaodingBallsP2")
    assert (bd[:, :2] >= 0.03).all() and (bd[:, :2] <= 0.3).all() and np.ptp(bd[:, 0]) > 0.1
    assert (bd[:, 8:] >= 0.018).all() and (bd[:, 8:] <= 0.024).all()
    assert (np.abs(bd[:, 2] - 1.0) <= 0.2).all() and (np.abs(bd[:, 3] - 0.005) <= 0.001).all() and (np.abs(bd[:, 4] - 1e-4) <= 2e-5).all()
    assert (td[:, 2] >= 0.02).all() and (td[:, 2] <= 0.03).all() and (td[:, 4] >= 4).all() and (td[:, 4] <= 6).all()
    assert np.allclose(td[:, 1], td[:, 0] - np.pi) and (td[:, 0] >= 0).all() and (td[:, 0] <= 2 * np.pi).all()
    assert np.allclose(qp, init)
    # P2 RSI: one zero-action step, balls teleported to the targets' xy, hand back at init,
    # counter stays 1, act keeps the post-step activation (baoding.py:610-638)
    obs, qp, qv, ac, tt, ti, td, bd = run("CustomMyoBaodingBallsP2", enable_rsi=True, rsi_probability=1.0, task_choice="fixed")
    assert (ti[:, 1] == 1).all() and np.allclose(tt, 0.02) and (ac > 0).all() and not qv.any()
    assert np.allclose(qp[:, :23], init[:23]) and np.allclose(qp[:, [25, 32]], init[[25, 32]])
    assert np.abs(qp[:, [23, 24]] - obs[:, 35:37]).max() < 5e-3 and np.abs(qp[:, [30, 31]] - obs[:, 38:40]).max() < 5e-3
    # determinism by (seed, env, episode); different seeds differ
    a = run("CustomMyoBaodingBallsP2", seed=9)[6]; bb = run("CustomMyoBaodingBallsP2", seed=9)[6]; c = run("CustomMyoBaodingBallsP2", seed=10)[6]
    assert np.array_equal(a, bb) and not np.array_equal(a, c)
    # beta-distributed options stay inside their ranges
    obs, qp, qv, ac, tt, ti, td, bd = run("CustomMyoBaodingBallsP2", beta_ball_mass=[0.5, 0.5], beta_ball_size=[2, 5],
                                         limit_init_angle=0.5, beta_init_angle=[0.9, 0.9])
    assert (bd[:, :2] >= 0.03).all() and (bd[:, :2] <= 0.3).all() and (bd[:, 8:] >= 0.018).all() and (bd[:, 8:] <= 0.024).all()
    assert (td[:, 0] >= 3 * np.pi / 4 - np.pi - 1e-9).all() and (td[:, 0] <= 3 * np.pi / 4 + np.pi + 1e-9).all()


def case_reset_goldens(lib, models, golden_dir, dtype):
    """tests/golden/reset_logic_goldens.json records what the REFERENCE reset() did on a recording
    fake (tools/make_golden.py).  For every archived curriculum config the device reset must
    touch exactly the same qpos slots, keep the untouched ones at init_qpos, draw the task
    parameters from the same ranges and take the RSI branch (one zero-action step) when the
    reference does with probability 0 / 1."""
    import json
    import os
    from myochallenge_amd.envs.config import make_task_cfg
    from myochallenge_amd.model import compile_model
    from helpers import Mem
    cases = json.load(open(os.path.join(golden_dir, "reset_logic_goldens.json")))
    cm = compile_model(models["hand"])
    mem = Mem(lib)
    init = models["hand"].qpos0.copy(); init[:23] = 0; init[0] = -1.57
    by_cfg = {}
    for c in cases:
        by_cfg.setdefault((c["variant"], c["config_index"]), []).append(c)
    checked = 0
    for (variant, ci), group in by_cfg.items():
        cfg = dict(group[0]["config"])
        name = "CustomMyoBaodingBallsP1" if variant == "p1" else "CustomMyoBaodingBallsP2"
        # reference side: union over seeds of the slots its final set_state/robot.reset changed
        ref_changed = np.zeros(37, bool)
        ref_rsi = []
        for c in group:
            finals = [k for k in c["calls"] if k["call"] in ("set_state", "robot.reset")]
            q = np.array(finals[-1]["qpos"])
            ginit = np.array(c["calls"][[k["call"] for k in c["calls"]].index("robot.reset")]["qpos"])
            ref_changed |= np.abs(q - ginit) > 0
            ref_rsi.append(any(k["call"] == "step" for k in c["calls"]))
            assert all(np.allclose(k["action"], 0) for k in c["calls"] if k["call"] == "step")   # RSI steps with zeros(39)
        n = 48
        b = native.Batch(native.Model(cm, lib), make_task_cfg(name, cm, **cfg), n, 0, 3, dtype)
        obs = mem.zeros((n, 86), np.float32); b.reset(None, obs)
        qp, tt = mem.zeros((n, 37)), mem.zeros(n)
        ti, td, bd = mem.zeros((n, 2), np.int32), mem.zeros((n, 9)), mem.zeros((n, 10))
        b.get_state(qp, None, None, tt); b.get_task(ti, td, bd)
        dev_changed = np.abs(mem.host(qp) - init).max(0) > 1e-12
        # ball xy slots move when RSI fires (targets differ from the init ball xy); the reference fake
        # returns a random obs, so those 4 slots are marked changed there as well
        assert (dev_changed == ref_changed).all(), (variant, ci, np.where(dev_changed != ref_changed))
        dev_rsi = mem.host(ti)[:, 1] == 1
        p = float(cfg.get("rsi_probability", 1)) if cfg.get("enable_rsi") else 0.0
        if p in (0.0, 1.0):
            assert all(r == bool(p) for r in ref_rsi) and (dev_rsi == bool(p)).all()
        else:
            assert abs(dev_rsi.mean() - p) < 0.25
        # parameter ranges the reference sampled from
        gx, gy, gp = cfg["goal_xrange"], cfg["goal_yrange"], cfg["goal_time_period"]
        t = mem.host(td)
        assert (t[:, 2] >= gx[0] - 1e-12).all() and (t[:, 2] <= gx[1] + 1e-12).all()
        assert (t[:, 3] >= gy[0] - 1e-12).all() and (t[:, 3] <= gy[1] + 1e-12).all()
        assert (t[:, 4] >= gp[0] - 1e-12).all() and (t[:, 4] <= gp[1] + 1e-12).all()
        for c in group:
            assert gx[0] - 1e-12 <= c["x_radius"] <= gx[1] + 1e-12 and gy[0] - 1e-12 <= c["y_radius"] <= gy[1] + 1e-12
        b.close(); checked += 1
    assert checked >= 20


def p2_ball_params(n, seed=0):
    """ball_d rows drawn from CustomBaodingP2Env's registration ranges (src/envs/__init__.py:62-73)."""
    rng = np.random.RandomState(seed)
    bd = np.zeros((n, 10))
    bd[:, 0:2] = rng.uniform(0.03, 0.3, (n, 2))
    for k, (nom, ch) in enumerate(zip((1.0, 0.005, 1e-4), (0.2, 0.001, 2e-5))):
        bd[:, 2 + k] = rng.uniform(nom - ch, nom + ch, n)
        bd[:, 5 + k] = rng.uniform(nom - ch, nom + ch, n)
    bd[:, 8:10] = rng.uniform(0.018, 0.024, (n, 2))
    return bd


def case_p2_ball_physics(lib, mj, dtype, tol, nsteps=25, n=6):
    """Config C's physics: per-env ball mass / friction / size (what CustomBaodingP2Env.reset writes into its
    model, baoding.py:559-604, with inertia / invweight / rbound left nominal) injected through
    myo_batch_set_task on the device and through OracleData.set_ball_params on the oracle."""
    bd = p2_ball_params(n)
    r = episode_drift(lib, mj, dtype, [(0.2, s) for s in range(n)], nsteps, ball_d=bd)
    assert r["err_qpos_rel"].max() <= tol and r["err_obs_abs"].max() <= max(tol, 1e-7), (r["err_qpos_rel"].max(1), r["err_obs_abs"].max(1))
    nominal = episode_drift(lib, mj, dtype, [(0.2, 0)], 5)      # and the parameters do reach the physics
    return r, nominal


def case_step_inner(lib, mj, dtype, tol, nsteps=12):
    """myo_batch_step_inner (the unwrapped env.step MixtureModelBaodingEnv.reset takes with its base policy,
    /root/reference/src/envs/baoding.py:700-711): masked envs advance exactly as the oracle's env step does — no TimeLimit,
    no Monitor accounting, no auto-reset — and the other envs are not touched."""
    mem = Mem(lib)
    cm, om, _ = oracle_for(mj)
    n = 6
    b = native.Batch(native.Model(cm, lib), make_task_cfg("CustomMyoBaodingBallsP1", cm, max_episode_steps=5), n, 0, 9, dtype)
    obs = mem.zeros((n, 86), np.float32)
    b.reset(None, obs)
    mask = np.array([1, 0, 1, 1, 0, 1], np.uint8)
    ocfg = make_cfg(task_ids(cm))
    orc = []
    for e in range(n):
        d = OracleData(om)
        d.reset(); d.qpos[:23] = 0; d.qpos[0] = -1.57
        orc.append((d, default_state()))
    qp0 = mem.zeros((n, om.nq)); b.get_state(qp0); qp0 = mem.host(qp0).copy()
    rng = np.random.RandomState(3)
    done, qp = mem.zeros(n, np.uint8), mem.zeros((n, om.nq))
    out = mem.arr(np.full((n, 86), 7.0), np.float32)
    for t in range(nsteps):                                   # more steps than max_episode_steps: no truncation in here
        a = np.clip(rng.normal(0, 0.3, (n, 39)), -1, 1).astype(np.float32)
        b.step_inner(mem.arr(mask, np.uint8), mem.arr(a, np.float32), out, done)
        b.get_state(qp)
        h_out, h_done, h_qp = mem.host(out), mem.host(done), mem.host(qp)
        for e in range(n):
            if not mask[e]:
                assert np.array_equal(h_qp[e], qp0[e]) and (h_out[e] == 7.0).all()        # untouched
                continue
            d, st = orc[e]
            o, c = baoding_step(d, ocfg, st, a[e])
            assert np.abs(h_out[e] - o).max() < max(tol, 2e-7), (t, e, np.abs(h_out[e] - o).max())
            assert rel_err(h_qp[e], d.qpos) < tol and bool(h_done[e]) == bool(c[6])
    b.close()


# ---------------------------------------------------------------- die reorient
def reorient_oracle_cfg(cm, tcfg):
    from myochallenge_amd.envs.reorient import reorient_ids
    from oracle.oracle import ReorientCfg
    ids, c = reorient_ids(cm), ReorientCfg()
    c.frame_skip, c.n_hand = tcfg.frame_skip, tcfg.n_hand
    c.object_sid, c.goal_sid, c.object_bid, c.gid0, c.gidn = (ids[k] for k in ("object_sid", "goal_sid", "object_bid", "object_gid0", "object_gidn"))
    c.drop_th, c.pos_th, c.rot_th = tcfg.drop_th, tcfg.ro_pos_th, tcfg.ro_rot_th
    for i in range(3):
        c.goal_obj_offset[i] = tcfg.ro_goal_obj_offset[i]
    for i in range(9):
        c.w[i] = tcfg.ro_weights[i]
    return c


def case_reorient(lib, dtype, tol, env_name="CustomMyoReorientP2", n=6, nsteps=30, horizon=12, seed=11, **kw):
    """CustomReorientEnv.step / reset on the device (task kind MYO_TASK_REORIENT) against the oracle's orc_reorient_step on twins
    that share each episode's draws: observation, the reward dictionary with its shaping terms, drop termination, TimeLimit
    truncation with the terminal observation, Monitor numbers, and at every reset the oracle's own reset tail
    (orc_reorient_reset_dists) on the device's post-reset state.  Env 0 starts with the die pushed off the palm (drop)."""
    from myochallenge_amd.envs.reorient import make_reorient_cfg
    from myochallenge_amd.synth_hand import synthetic_hand_die
    from oracle.oracle import ReorientState, reorient_reset_dists, reorient_set_die, reorient_step
    mem = Mem(lib)
    cm, om, _ = oracle_for(synthetic_hand_die())
    tcfg = make_reorient_cfg(env_name, cm, max_episode_steps=horizon, **kw)
    ocfg = reorient_oracle_cfg(cm, tcfg)
    b = native.Batch(native.Model(cm, lib), tcfg, n, 0, seed, dtype)
    nobs, ng = b.obs_dim, ocfg.gidn - ocfg.gid0
    assert nobs == 2 * tcfg.n_hand + 18 + om.na
    obs, rew, done, trunc = mem.zeros((n, nobs), np.float32), mem.zeros(n, np.float32), mem.zeros(n, np.uint8), mem.zeros(n, np.uint8)
    term, comps, ep = mem.zeros((n, nobs), np.float32), mem.zeros((n, 8), np.float32), mem.zeros((n, 2), np.float32)
    qp, qv, ac, tm = mem.zeros((n, om.nq)), mem.zeros((n, om.nv)), mem.zeros((n, om.na)), mem.zeros(n)
    ti, td, bd, fr = mem.zeros((n, 2), np.int32), mem.zeros((n, 9)), mem.zeros((n, 10)), mem.zeros((n, ng, 3))
    b.reset(None, obs)
    b.get_state(qp, qv, ac, tm)
    q = mem.host(qp).copy()
    q[0, -7] += 0.25                                 # env 0: the die starts beside the palm and falls
    b.set_state(mem.arr(q), None, None, None)
    twins = [None] * n
    obs_tol = max(tol, 2e-7)                         # the device returns float32 observations (angles reach 2 pi: relative)

    def sync(e, check_obs=None):
        """oracle twin of env e from the device's state and episode draws; runs the oracle's reset tail on it"""
        b.get_state(qp, qv, ac, tm); b.get_task(ti, td, bd); b.object_friction(None, fr)
        h = [mem.host(x) for x in (qp, qv, ac, tm, td, bd, fr)]
        d = OracleData(om)
        d.reset()
        d.qpos[:], d.qvel[:], d.act[:] = h[0][e], h[1][e], h[2][e]
        d.arr("time")[0] = h[3][e]
        st = ReorientState()
        for k in range(3):
            st.goal_pos[k] = h[4][e, k]
        for k in range(4):
            st.goal_quat[k] = h[4][e, 3 + k]
        reorient_set_die(d, ocfg, h[6][e], h[5][e, 8])
        o = reorient_reset_dists(d, ocfg, st)
        if check_obs is not None:                    # a reset just happened on the device: same observation, same distances
            assert (np.abs(check_obs - o) <= obs_tol * np.maximum(1, np.abs(o))).all(), np.abs(check_obs - o).max()
            assert abs(st.pos_dist - h[4][e, 7]) < 1e-12 + tol and abs(st.rot_dist - h[4][e, 8]) < 1e-12 + tol
            assert np.all(d.qpos[1:tcfg.n_hand] == 0) and d.qpos[0] == -1.5 and np.all(d.qvel == 0) and d.arr("time")[0] == 0
        else:
            st.pos_dist, st.rot_dist = h[4][e, 7], h[4][e, 8]
        twins[e] = (d, st)
        return h

    h0 = None
    for e in range(n):
        h0 = sync(e, None if e == 0 else mem.host(obs)[e])
    rng = np.random.RandomState(5)
    rets, lens = np.zeros(n), np.zeros(n, int)
    seen = dict(drop=0, trunc=0)
    for t in range(nsteps):
        a = np.clip(rng.normal(0, 0.4, (n, om.nu)), -1, 1).astype(np.float32)
        b.step(mem.arr(a, np.float32), obs, rew, done, trunc, term, comps, ep)
        ho, hr, hd, ht, hterm, hc, hep = (mem.host(x).copy() for x in (obs, rew, done, trunc, term, comps, ep))
        for e in range(n):
            d, st = twins[e]
            o, c = reorient_step(d, ocfg, st, a[e])
            rets[e] += c[9]; lens[e] += 1
            dev_obs = hterm[e] if hd[e] else ho[e]
            assert (np.abs(dev_obs - o) <= obs_tol * np.maximum(1, np.abs(o))).all(), (t, e, np.abs(dev_obs - o).max())
            want = np.array([c[0], c[1], c[5], c[4], c[6], c[7], c[8], c[9]])        # comps = pos_dist, rot_dist, act_reg, alive, sparse, solved, done, dense
            assert np.abs(hc[e] - want).max() < max(tol, 1e-6) * (1 + np.abs(want).max()), (t, e, hc[e], want)
            assert abs(hr[e] - c[9]) < max(tol, 1e-6) * (1 + abs(c[9]))
            drop, timeout = bool(c[8]), lens[e] >= horizon
            assert bool(hd[e]) == (drop or timeout) and bool(ht[e]) == (timeout and not drop)
            if hd[e]:
                seen["drop" if drop else "trunc"] += 1
                assert abs(hep[e, 0] - rets[e]) < 1e-4 * (1 + abs(rets[e])) and int(hep[e, 1]) == lens[e]
                rets[e], lens[e] = 0, 0
                sync(e, ho[e])
    b.close()
    assert seen["drop"] >= 1 and seen["trunc"] >= n - 1, seen
    return seen


def case_bad_state(lib, mj, dtype):
    """A numerically blown-up env (MuJoCo: mj_checkPos / mj_checkVel / mj_checkAcc warn and reset the data) is not an error of
    the batch: it ends its episode with done = 1, reward 0, zero reward components except `done`, finite terminal and returned
    observations (the reset observation), bad_state = 1 — and the other envs of the batch do not notice."""
    mem = Mem(lib)
    cm, om, _ = oracle_for(mj)
    n = 4
    mk = lambda: native.Batch(native.Model(cm, lib), make_task_cfg("CustomMyoBaodingBallsP1", cm), n, 0, 5, dtype)
    a_b, b_b = mk(), mk()                       # b is the clean twin
    outs = []
    for b in (a_b, b_b):
        obs, rew, done, trunc = mem.zeros((n, 86), np.float32), mem.zeros(n, np.float32), mem.zeros(n, np.uint8), mem.zeros(n, np.uint8)
        term, comps, ep, bad = mem.zeros((n, 86), np.float32), mem.zeros((n, 8), np.float32), mem.zeros((n, 2), np.float32), mem.arr(np.full(n, 7), np.uint8)
        b.reset(None, obs)
        b.set_bad_state_buffer(bad)
        outs.append((obs, rew, done, trunc, term, comps, ep, bad))
    qv = mem.zeros((n, om.nv)); a_b.get_state(None, qv)
    h = mem.host(qv).copy(); h[1, 3] = 1e12; h[2, 30] = np.nan          # env 1: absurd finger velocity, env 2: NaN ball velocity
    a_b.set_state(None, mem.arr(h), None, None)
    rng = np.random.RandomState(0)
    for t in range(3):
        act = mem.arr(np.clip(rng.normal(0, 0.2, (n, 39)), -1, 1), np.float32)
        for b, o in ((a_b, outs[0]), (b_b, outs[1])):
            b.step(act, *o[:7])
        A, B = [[mem.host(x).copy() for x in o] for o in outs]
        for k in range(7):
            assert np.isfinite(A[k]).all(), (t, k)
        if t == 0:
            assert list(A[7]) == [0, 1, 1, 0] and list(A[2]) == [0, 1, 1, 0] and list(A[3]) == [0, 0, 0, 0]
            for e in (1, 2):
                assert A[1][e] == 0 and np.array_equal(A[5][e], [0, 0, 0, 0, 0, 0, 1, 0])
                assert np.array_equal(A[4][e], A[0][e])                                               # terminal = returned observation ...
                assert abs(A[0][e][0] + 1.57) < 1e-6 and np.abs(A[0][e][1:23]).max() == 0 and np.abs(A[0][e][47:]).max() == 0   # ... of a fresh episode
                assert A[6][e][1] == 1                                                                       # Monitor: an episode of one step
        else:
            assert list(A[7]) == [0, 0, 0, 0]
        for e in (0, 3):                                     # the healthy envs are bit-identical to the clean twin's
            for k in range(7):
                assert np.array_equal(A[k][e], B[k][e]), (t, e, k)
    a_b.close(); b_b.close()


def reorient_drift(lib, dtype, n=16, nsteps=150, horizon=150, env_name="CustomMyoReorientP2", seed=11, sigma=0.2, local=False, **kw):
    """Whole-episode drift record for the die-reorient env (BASELINE config E physics + task layer): `n` envs, `nsteps` env steps
    (x frame_skip 5 substeps) with auto-reset, device vs oracle twins that share each episode's draws (as case_reorient, which
    ASSERTS per-step bounds on short episodes; this one MEASURES).  err_q = max |qpos - qpos_oracle| / max |qpos_oracle|,
    err_obs = max |obs - obs_oracle| / max(1, |obs_oracle|) (the observation carries Euler angles up to 2 pi).  A stream whose
    two sides end an episode on different steps reports 1.0 from there on.
    local=True: the LOCAL error instead (as local_error does for Baoding) — before every env step the device is put on its twin's
    state (qpos, qvel, act, time, warm start), so err[t] is what one env step of the stepper adds; an episode-end disagreement
    only voids that step (the device env is reset by hand and both sides go on)."""
    from myochallenge_amd.envs.reorient import make_reorient_cfg
    from myochallenge_amd.synth_hand import synthetic_hand_die
    from oracle.oracle import ReorientState, reorient_reset_dists, reorient_set_die, reorient_step
    mem = Mem(lib)
    cm, om, _ = oracle_for(synthetic_hand_die())
    tcfg = make_reorient_cfg(env_name, cm, max_episode_steps=horizon, **kw)
    ocfg = reorient_oracle_cfg(cm, tcfg)
    b = native.Batch(native.Model(cm, lib), tcfg, n, 0, seed, dtype)
    nobs, ng = b.obs_dim, ocfg.gidn - ocfg.gid0
    obs, rew, done, trunc = mem.zeros((n, nobs), np.float32), mem.zeros(n, np.float32), mem.zeros(n, np.uint8), mem.zeros(n, np.uint8)
    term = mem.zeros((n, nobs), np.float32)
    qp, qv, ac, tm = mem.zeros((n, om.nq)), mem.zeros((n, om.nv)), mem.zeros((n, om.na)), mem.zeros(n)
    ti, td, bd, fr, ws = mem.zeros((n, 2), np.int32), mem.zeros((n, 9)), mem.zeros((n, 10)), mem.zeros((n, ng, 3)), mem.zeros((n, om.nv))
    b.reset(None, obs)

    def twin(e):
        b.get_state(qp, qv, ac, tm); b.get_task(ti, td, bd); b.object_friction(None, fr); b.warmstart(ws, None)
        h = [mem.host(x) for x in (qp, qv, ac, tm, td, bd, fr, ws)]
        d = OracleData(om)
        d.reset()
        d.qpos[:], d.qvel[:], d.act[:] = h[0][e], h[1][e], h[2][e]
        d.arr("time")[0] = h[3][e]
        d.arr("qacc_warmstart")[:] = h[7][e]
        st = ReorientState()
        for k in range(3):
            st.goal_pos[k] = h[4][e, k]
        for k in range(4):
            st.goal_quat[k] = h[4][e, 3 + k]
        reorient_set_die(d, ocfg, h[6][e], h[5][e, 8])
        reorient_reset_dists(d, ocfg, st)
        return [d, st, 0]

    twins = [twin(e) for e in range(n)]
    rngs = [np.random.RandomState(100 + e) for e in range(n)]
    err_q, err_obs, ends, split = np.zeros((n, nsteps)), np.zeros((n, nsteps)), [[] for _ in range(n)], [None] * n
    for t in range(nsteps):
        a = np.stack([np.clip(r.normal(0, sigma, om.nu), -1, 1) for r in rngs]).astype(np.float32)
        if local:
            b.set_state(mem.arr(np.stack([tw[0].qpos for tw in twins])), mem.arr(np.stack([tw[0].qvel for tw in twins])),
                        mem.arr(np.stack([tw[0].act for tw in twins])), mem.arr(np.array([tw[0].arr("time")[0] for tw in twins])))
            b.warmstart(None, mem.arr(np.stack([np.array(tw[0].arr("qacc_warmstart")) for tw in twins])))
        b.step(mem.arr(a, np.float32), obs, rew, done, trunc, term)
        b.get_state(qp)
        ho, hd, ht, hterm, hq = (mem.host(x).copy() for x in (obs, done, trunc, term, qp))
        for e in range(n):
            if local and split[e] is not None:           # (a voided step: both sides restart below)
                split[e] = None
            if split[e] is not None:
                err_q[e, t] = err_obs[e, t] = 1.0
                continue
            d, st, el = twins[e]
            o, c = reorient_step(d, ocfg, st, a[e])
            el += 1
            o_done = bool(c[8]) or el >= horizon
            if bool(hd[e]) != o_done:
                split[e] = t
                err_q[e, t] = err_obs[e, t] = 1.0
                if local:                                # put both sides on a fresh episode of the device's
                    if not bool(hd[e]):
                        mk = np.zeros(n, np.uint8); mk[e] = 1
                        b.reset(mem.arr(mk, np.uint8), obs)
                    ends[e].append(t)
                    twins[e] = twin(e)
                continue
            dev_obs = hterm[e] if o_done else ho[e]
            err_obs[e, t] = (np.abs(dev_obs - o) / np.maximum(1, np.abs(o))).max()
            if o_done:
                err_q[e, t] = err_q[e, t - 1] if t else 0.0
                ends[e].append(t)
                twins[e] = twin(e)
            else:
                err_q[e, t] = np.abs(hq[e] - d.qpos).max() / np.abs(d.qpos).max()
                twins[e][2] = el
    b.close()
    return {"streams": [[sigma, 100 + e] for e in range(n)], "err_qpos_rel": err_q, "err_obs_abs": err_obs, "episode_ends": ends,
            "episode_end_disagreement_at": split}
