"""Pins for the oracle's UNPINNED stages from published closed forms.  No MuJoCo output exists in this image
(SURVEY.md §8c), so these do not make the stepping parity "pinned" — they narrow the room for a shared misreading: every expected
value below is derived in this file from the formulas MuJoCo 2.1 documents (computation chapter: solver parameters, impedance,
pyramidal cones; modeling chapter: muscle actuators), not taken from oracle/ or csrc/.

* impedance d(r): table points of the documented sigmoid (solimp = d0, dwidth, width, midpoint, power);
* reference acceleration / regulariser of a contact row at prescribed penetrations:  aref = -b v - k d(r) r,  k = 1/(dmax^2 tc^2 zeta^2),
  b = 2/(dmax tc),  R = (1 - d)/d * diagApprox,  pyramidal rows R_py = 2 mu^2 R;
* a sphere resting on a plane: the penetration at which the four pyramid rows carry the weight, 4 D k d(r) r = m g;
* muscle force-length / force-velocity / passive curves at their documented knots.
"""
import numpy as np
import pytest

from helpers import oracle_for
from myochallenge_amd.model import compile_model
from myochallenge_amd.setconst import set_const
from myochallenge_amd.synth_hand import _Builder
from oracle.oracle import OracleData, OracleModel

SOLREF, SOLIMP = (0.02, 1.0), (0.9, 0.95, 0.001, 0.5, 2.0)       # MuJoCo's defaults
MASS, RADIUS, MU, G = 0.3, 0.05, 1.0, 9.81


def impedance(r, solimp=SOLIMP):
    """MuJoCo docs, "Solver parameters": d(r) rises from d0 at r = 0 to dwidth at r >= width along a sigmoid made of two power
    functions joined at `midpoint`."""
    d0, dw, width, mid, p = solimp
    x = min(1.0, abs(r) / width)
    y = x ** p / mid ** (p - 1) if x <= mid else 1.0 - (1.0 - x) ** p / (1.0 - mid) ** (p - 1)
    return d0 + y * (dw - d0)


def sphere_on_plane():
    B = _Builder()
    B.add_geom("plane", 0, 0, (0, 0, 0), collide=1)
    b = B.add_body("ball", 0, (0.0, 0.0, RADIUS), mass=MASS, inertia=(3e-4, 3e-4, 3e-4))
    B.add_joint("ball_free", b, 0, damping=0.0, armature=0.0)
    B.add_geom("ball", b, 2, (RADIUS,), collide=1, friction=(MU, 0.005, 0.0001))
    m = B.finish()
    m.arrays["geom_contype"][:] = 1
    m.arrays["geom_conaffinity"][:] = 1
    m.arrays["geom_margin"][:] = 0.0
    m.arrays["geom_solref"][:] = SOLREF
    m.arrays["geom_solimp"][:] = SOLIMP
    m.arrays["dof_damping"][:] = 0.0
    set_const(m)
    return m


def row_constants(r):
    """k, b, d, D of one pyramid row of the sphere-plane contact at penetration r > 0 (impratio 1)"""
    d0, dmax = SOLIMP[0], SOLIMP[1]
    tc, zeta = SOLREF
    k, b = 1.0 / (dmax * dmax * tc * tc * zeta * zeta), 2.0 / (dmax * tc)
    d = impedance(r)
    diag = (1.0 / MASS) * (1.0 + MU * MU)          # diagApprox of a translational pyramid row: tran + mu^2 tran, tran = 1/m (free body) + 0 (world)
    R = (1.0 - d) / d * diag
    return k, b, d, 1.0 / (2.0 * MU * MU * R)


def test_impedance_table_points():
    w = SOLIMP[2]
    for r, want in ((0.0, 0.9), (0.25 * w, 0.90625), (0.5 * w, 0.925), (0.75 * w, 0.94375), (w, 0.95), (3 * w, 0.95)):
        assert abs(impedance(r) - want) < 1e-15


@pytest.mark.parametrize("frac", [0.1, 0.25, 0.5, 0.75, 1.0, 2.5])
def test_contact_row_reference_and_regulariser(frac):
    """oracle rows of a sphere pressed `frac * width` into a plane, moving down at 0.1 m/s"""
    m = sphere_on_plane()
    cm = compile_model(m)
    d = OracleData(OracleModel(cm.to_blob()))
    r, v = frac * SOLIMP[2], -0.1
    d.qpos[:] = [0, 0, RADIUS - r, 1, 0, 0, 0]
    d.qvel[:] = [0, 0, v, 0, 0, 0]
    d.forward()
    assert d.ncon == 1 and d.nefc == 4
    k, b, imp, D = row_constants(r)
    aref = -b * v + k * imp * r                              # -b (J v) - k d (pos - margin), pos = -r, J v = v (normal row component; the tangential part is 0)
    assert np.allclose(np.array(d.efc_aref)[:4], aref, rtol=1e-12, atol=0)
    assert np.allclose(np.array(d.efc_D)[:4], D, rtol=1e-12, atol=0)
    assert np.allclose(np.array(d.efc_pos)[:4], -r, rtol=1e-12, atol=1e-18)


def test_sphere_rests_at_the_closed_form_penetration():
    m = sphere_on_plane()
    cm = compile_model(m)
    d = OracleData(OracleModel(cm.to_blob()))
    d.qpos[:] = [0, 0, RADIUS + 1e-4, 1, 0, 0, 0]
    for _ in range(4000):                                     # 8 s at 2 ms: critically damped contact, settles to rounding
        d.step()
    assert np.abs(np.array(d.qvel)).max() < 1e-10 and d.ncon == 1
    r_oracle = RADIUS - float(d.qpos[2])
    # fixed point of  4 D(r) k d(r) r = m g  (every pyramid row carries a quarter of the weight: f_i = -D jar_i, jar_i = -aref_i at rest)
    r = 1e-4
    for _ in range(200):
        k, _, imp, D = row_constants(r)
        r = MASS * G / (4.0 * D * k * imp)
    assert 1e-5 < r < SOLIMP[2]
    assert abs(r_oracle - r) <= 1e-9 * r, (r_oracle, r)
    # and the contact force it reports is the weight
    f = np.array(d.efc_force)[:4]
    assert abs(f.sum() - MASS * G) <= 1e-9 * MASS * G and np.allclose(f, f[0], rtol=1e-9)


# ---- muscle curves (MuJoCo docs, "Muscle actuators": FL is a bell of four quadratic pieces with knots lmin, a = (lmin+1)/2, 1,
# b = (1+lmax)/2, lmax; FV is 0 at V = -1, 1 at V = 0 and saturates at fvmax from V = fvmax - 1 on; FP is 0 up to L = 1, fpmax/2 at b
# and linear beyond) on the one-muscle "load" model: length = 0.1 - q, gainprm = range (0.75, 1.05), force -1 -> F0 = scale / acc0
def _load(models):
    cm, om, d = oracle_for(models["load"])
    mj = models["load"]
    g = mj.actuator_gainprm[0]
    lr = mj.actuator_lengthrange[0]
    L0 = (lr[1] - lr[0]) / (g[1] - g[0])
    F0 = g[3] / mj.actuator_acc0[0]
    return d, dict(r0=g[0], lr0=lr[0], L0=L0, F0=F0, lmin=g[4], lmax=g[5], vmax=g[6], fpmax=g[7], fvmax=g[8])


def _force(d, p, L, V, act):
    length = p["lr0"] + (L - p["r0"]) * p["L0"]
    d.qpos[:] = [0.1 - length]
    d.qvel[:] = [-V * p["vmax"] * p["L0"]]                   # length = 0.1 - q: d length / dt = -qvel
    d.act[:] = act
    d.ctrl[:] = act
    d.forward()
    assert abs(float(np.array(d.actuator_length)[0]) - length) < 1e-14
    return float(np.array(d.actuator_force)[0])


def test_muscle_curve_knots(models):
    d, p = _load(models)
    F0, lmin, lmax, fvmax, fpmax = p["F0"], p["lmin"], p["lmax"], p["fvmax"], p["fpmax"]
    a, b = 0.5 * (lmin + 1.0), 0.5 * (1.0 + lmax)
    FP = lambda L: 0.0 if L <= 1 else (0.5 * ((L - 1) / (b - 1)) ** 2 * fpmax if L <= b else fpmax * (0.5 + (L - b) / (b - 1)))
    # force-length at V = 0, full activation: FL = 0, 1/2, 1, 1/2, 0 at lmin, a, 1, b, lmax (passive part added beyond L = 1)
    for L, FL in ((lmin, 0.0), (a, 0.5), (1.0, 1.0), (b, 0.5), (lmax, 0.0), (0.5 * (a + 1), 0.875)):
        assert abs(_force(d, p, L, 0.0, 1.0) + F0 * (FL + FP(L))) <= 1e-12 * F0, L
    # force-velocity at L = 1: 0 at V = -1, (V + 1)^2 below 0, 1 at 0, fvmax - (y - V)^2 / y up to y = fvmax - 1, fvmax beyond
    y = fvmax - 1.0
    for V, FV in ((-1.5, 0.0), (-1.0, 0.0), (-0.5, 0.25), (0.0, 1.0), (0.5 * y, fvmax - 0.25 * y), (y, fvmax), (2.0, fvmax)):
        assert abs(_force(d, p, 1.0, V, 1.0) + F0 * FV) <= 1e-12 * F0, V
    # passive force alone (activation 0)
    for L in (0.9, 1.0, 0.5 * (1 + b), b, b + 0.1):
        assert abs(_force(d, p, L, 0.0, 0.0) + F0 * FP(L)) <= 1e-12 * F0, L


def test_muscle_activation_dynamics_time_constants(models):
    """act_dot = (ctrl - act) / tau,  tau = tau_act (0.5 + 1.5 act) when ctrl > act, tau_deact / (0.5 + 1.5 act) otherwise"""
    d, p = _load(models)
    tau_a, tau_d = models["load"].actuator_dynprm[0][:2]
    for act, ctrl in ((0.2, 0.9), (0.9, 0.2), (0.0, 1.0), (1.0, 0.0), (0.5, 0.5)):
        d.qpos[:] = [0.0]; d.qvel[:] = [0.0]; d.act[:] = act; d.ctrl[:] = ctrl
        d.forward()
        tau = tau_a * (0.5 + 1.5 * act) if ctrl > act else tau_d / (0.5 + 1.5 * act)
        assert abs(float(np.array(d.act_dot)[0]) - (ctrl - act) / tau) <= 1e-12 * max(1.0, abs((ctrl - act) / tau))


def test_the_two_newton_line_searches_of_the_oracle_agree():
    """VERDICT r04 item 10.  The oracle's default line search (safeguarded Newton on p'(alpha)) and the bracketing structure of MuJoCo 2.1's
    PrimalSearch restated behind a switch (myo_oracle.c: primal_search; p0 / p1 initialisation, one-sided steps to a sign change, three
    candidates per iteration) stop inside the same gradient tolerance: over env steps of the bench workload taken from the same state
    they must leave qpos within 1e-10 of each other (measured: 2e-16 per env step, 4e-12 over whole episodes of 200 steps, profiles/r05_oracle_linesearch.json) — so a
    future comparison with true MuJoCo trajectories will not trip over the line search first.  Stepping parity stays UNPINNED."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import oracle_linesearch as ols
    r = ols.measure([(0.135, 0), (0.135, 3), (0.08, 5)], 30)
    assert max(r["local_qpos_rel"]) <= 1e-10 and max(r["episode_qpos_rel"]) <= 1e-9, r
    assert r["env_steps"] == 90
