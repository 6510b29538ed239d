"""Narrow phases beyond MuJoCo's sphere / capsule primitives: capsule-box, box-box
(vertex-face candidates), sphere / capsule against cylinder and ellipsoid, plane against ellipsoid and cylinder.

A "contact zoo" model — one free body per geom type over a ground plane, plus a static box — is put into seeded random
configurations of each pair (random orientations, separations from deep penetration to just apart), and the stepper
(lane-serial build on CPU, HIP kernels under -m gpu) is compared with the oracle through the C ABI: contact / constraint-row
counts equal, constraint reference accelerations, impedances and the solved accelerations to 1e-9 (fp64) / 1e-4 (mixed),
then a short trajectory of everything falling onto the plane and the static box.  The oracle's own geometry is checked
against brute-force sampling of the two solids (test_oracle_contact_geometry)."""
import numpy as np
import pytest

from helpers import Mem, forward_dump, rel_err
from myochallenge_amd import native
from myochallenge_amd.mathutil import quat_to_mat
from myochallenge_amd.model import compile_model
from myochallenge_amd.setconst import set_const
from myochallenge_amd.synth_hand import _Builder
from oracle.oracle import OracleData, OracleModel

SPH, CAP, ELL, CYL, BOX = 2, 3, 4, 5, 6
BODIES = ("sphere", "capsule", "box", "cylinder", "ellipsoid")          # free bodies, in qpos order
SIZES = {"sphere": (0.03,), "capsule": (0.02, 0.05), "box": (0.04, 0.03, 0.02), "cylinder": (0.03, 0.04), "ellipsoid": (0.05, 0.03, 0.02),
         "slab": (0.1, 0.1, 0.02)}
TYPES = {"sphere": SPH, "capsule": CAP, "box": BOX, "cylinder": CYL, "ellipsoid": ELL, "slab": BOX}
#             contype, conaffinity: collide iff (ct1 & ca2) | (ct2 & ca1); chosen so that exactly the supported pairs exist
BITS = {"plane": (1, 0), "sphere": (2, 1), "capsule": (4, 3), "cylinder": (8, 7), "ellipsoid": (16, 7), "box": (32, 6), "slab": (64, 36)}
SLAB_POS = np.array([0.0, 0.6, 0.5])      # (everything within ~1 m: the mixed stepper's fp32 stages work relative to body 1)
PAIRS = [("plane", "ellipsoid"), ("plane", "cylinder"), ("sphere", "cylinder"), ("sphere", "ellipsoid"), ("capsule", "cylinder"),
         ("capsule", "ellipsoid"), ("capsule", "box"), ("capsule", "slab"), ("box", "slab"), ("sphere", "box"), ("sphere", "capsule")]


def zoo_model(margin=0.002):
    B = _Builder()
    B.add_geom("plane", 0, 0, (0, 0, 0), collide=1)
    B.add_geom("slab", 0, BOX, SIZES["slab"], tuple(SLAB_POS), collide=1)
    for k, name in enumerate(BODIES):
        b = B.add_body(name, 0, (0.4 * k, 0.0, 1.0), mass=0.05, inertia=(2e-5, 3e-5, 4e-5))
        B.add_joint(name + "_free", b, 0)
        B.add_geom(name, b, TYPES[name], SIZES[name], collide=1)
    m = B.finish()
    for g, name in enumerate(m.names["geom"]):
        m.arrays["geom_contype"][g], m.arrays["geom_conaffinity"][g] = BITS[name]
        if m.arrays["geom_type"][g] == ELL:
            m.arrays["geom_rbound"][g] = max(SIZES["ellipsoid"])
    m.arrays["geom_margin"][:] = margin         # contacts switch on before touching: both signs of dist are exercised
    set_const(m)
    return m


def extent(name):
    s = SIZES[name]
    t = TYPES.get(name, 0)
    if t == SPH:
        return s[0], s[0]
    if t == CAP:
        return s[0], s[0] + s[1]
    if t == CYL:
        return min(s), float(np.hypot(s[0], s[1]))
    return min(s), float(np.linalg.norm(s)) if t == BOX else max(s)


def rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def configurations(m, seed=0, per_pair=8):
    """(pair, qpos) states: the two geoms of `pair` at a random relative pose, everything else parked far away"""
    rng = np.random.RandomState(seed)
    out = []
    for a, b in PAIRS:
        for k in range(per_pair):
            q = m.qpos0.copy()
            lo = 0.6 * (extent(a)[0] + extent(b)[0]) if a != "plane" else 0.6 * extent(b)[0]
            hi = 1.05 * (extent(a)[1] + extent(b)[1]) if a != "plane" else 1.05 * extent(b)[1]
            dist = rng.uniform(lo, hi)
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            if a == "plane":
                ib = BODIES.index(b)
                q[7 * ib:7 * ib + 3] = [0.4 * ib, 0.0, dist]
                q[7 * ib + 3:7 * ib + 7] = rand_quat(rng)
            elif b == "slab":
                ia = BODIES.index(a)
                q[7 * ia:7 * ia + 3] = SLAB_POS + u * dist
                q[7 * ia + 3:7 * ia + 7] = rand_quat(rng)
            else:
                ia, ib = BODIES.index(a), BODIES.index(b)
                c = np.array([0.2, -0.5, 0.7])
                q[7 * ia:7 * ia + 3] = c
                q[7 * ib:7 * ib + 3] = c + u * dist
                q[7 * ia + 3:7 * ia + 7] = rand_quat(rng)
                q[7 * ib + 3:7 * ib + 7] = rand_quat(rng)
            out.append(((a, b), q))
    return out


def case_zoo_forward(lib, dtype, tol):
    mem = Mem(lib)
    m = zoo_model()
    cm = compile_model(m)                       # unsupported_contacts="error": every colliding pair of the zoo has a narrow phase
    assert cm.dropped_pairs == []
    om = OracleModel(cm.to_blob())
    rng = np.random.RandomState(1)
    touched = {p: 0 for p in PAIRS}
    for pair, q in configurations(m):
        d = OracleData(om)
        v = rng.normal(0, 0.3, om.nv)
        d.qpos[:], d.qvel[:] = q, v
        d.forward()
        get, b = forward_dump(lib, mem, cm, q, v, np.zeros(0), np.zeros(0), dtype)
        cnt = get("counts", 4)
        assert (int(cnt[0]), int(cnt[1])) == (d.ncon, d.nefc), (pair, cnt, d.ncon, d.nefc)
        touched[pair] += d.ncon
        if d.nefc:
            for name in ("efc_aref", "efc_D"):
                ref = np.array(getattr(d, name))[:d.nefc]
                assert rel_err(get(name, d.nefc), ref) < tol, (pair, name, rel_err(get(name, d.nefc), ref))
        for name in ("qacc_smooth", "qacc"):
            ref = np.array(getattr(d, name))
            assert rel_err(get(name, ref.size), ref) < tol, (pair, name, rel_err(get(name, ref.size), ref))
        b.close()
    assert all(n > 0 for n in touched.values()), touched          # every pair type produced contacts in some configuration
    return touched


def case_zoo_drop(lib, dtype, tol, nsteps=150):
    """everything dropped from a few cm onto the plane / the static box: 150 substeps, trajectory against the oracle"""
    mem = Mem(lib)
    m = zoo_model()
    cm = compile_model(m)
    om = OracleModel(cm.to_blob())
    rng = np.random.RandomState(3)
    q = m.qpos0.copy()
    for k, name in enumerate(BODIES):
        q[7 * k:7 * k + 3] = [0.5 * k, 0.0, extent(name)[1] + 0.01] if name != "capsule" else SLAB_POS + [0.02, 0.01, 0.02 + extent(name)[1] + 0.01]
        q[7 * k + 3:7 * k + 7] = rand_quat(rng)
    n = 2
    b = native.Batch(native.Model(cm, lib), None, n, 0, 0, dtype)
    b.set_state(mem.arr(np.tile(q, (n, 1))), mem.zeros((n, om.nv)), mem.zeros((n, 0)), mem.zeros(n))
    d = OracleData(om)
    d.qpos[:] = q
    qp, qv = mem.zeros((n, om.nq)), mem.zeros((n, om.nv))
    ncon_seen = 0
    for i in range(nsteps):
        d.step()
        ncon_seen = max(ncon_seen, d.ncon)
        b.physics_step(None, 1)
        if i % 10 == 9:
            b.get_state(qp, qv)
            assert rel_err(mem.host(qp)[1], d.qpos) < tol, (i, rel_err(mem.host(qp)[1], d.qpos))
    assert ncon_seen >= 4 and not d.bad
    b.close()


def test_zoo_forward_on_emulation(emu_lib):
    t = case_zoo_forward(emu_lib, native.MYO_F64, 1e-9)
    assert t[("box", "slab")] >= 4                              # vertex-face candidates of the box-box pair
    case_zoo_forward(emu_lib, native.MYO_MIXED, 1e-4)


def test_zoo_drop_on_emulation(emu_lib):
    case_zoo_drop(emu_lib, native.MYO_F64, 1e-8)


@pytest.mark.gpu
def test_zoo_on_gpu(hip_lib):
    case_zoo_forward(hip_lib, native.MYO_F64, 1e-9)
    case_zoo_forward(hip_lib, native.MYO_MIXED, 1e-4)
    case_zoo_drop(hip_lib, native.MYO_F64, 1e-8)
    case_zoo_drop(hip_lib, native.MYO_MIXED, 1e-4)


# ------------------------------------------------------------------ the oracle's geometry against brute force
def _solid_points(name, n, rng):
    """points densely covering the SURFACE of the solid in its own frame"""
    s, t = SIZES[name], TYPES[name]
    if t == SPH:
        u = rng.normal(size=(n, 3)); return s[0] * u / np.linalg.norm(u, axis=1, keepdims=True)
    if t == ELL:
        u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True); return u * np.array(s)
    if t == BOX:
        p = rng.uniform(-1, 1, (n, 3)); k = rng.randint(0, 3, n); p[np.arange(n), k] = np.sign(p[np.arange(n), k]); return p * np.array(s)
    if t == CYL:
        th = rng.uniform(0, 2 * np.pi, n); side = rng.rand(n) < 0.6
        z = np.where(side, rng.uniform(-s[1], s[1], n), np.sign(rng.uniform(-1, 1, n)) * s[1])
        r = np.where(side, s[0], s[0] * np.sqrt(rng.rand(n)))
        return np.stack([r * np.cos(th), r * np.sin(th), z], 1)
    if t == CAP:
        u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        z = rng.uniform(-s[1], s[1], n) * (rng.rand(n) < 0.6)
        p = s[0] * u; side = z != 0
        p[side, 2] = 0; p[side] = s[0] * p[side] / np.maximum(1e-12, np.linalg.norm(p[side], axis=1, keepdims=True))
        p[:, 2] += np.where(side, z, np.sign(u[:, 2]) * s[1])
        return p
    raise ValueError(name)


def test_oracle_contact_geometry():
    """For separated configurations the oracle's contact distance must equal the true distance between the two solids (brute
    force over dense surface samples, to the sampling resolution), its normal must point from geom 1 to geom 2, and the contact
    position must lie between the surfaces."""
    m = zoo_model(margin=0.08)          # a wide margin: separated configurations produce a contact whose distance can be checked
    cm = compile_model(m)
    om = OracleModel(cm.to_blob())
    rng = np.random.RandomState(5)
    gname = m.names["geom"]
    checked = 0
    for pair, q in configurations(m, seed=7, per_pair=10):
        a, b = pair
        if a == "plane" or ("capsule" in pair and "ellipsoid" in pair):      # (capsule-ellipsoid uses the ellipsoid's own metric: not the Euclidean closest point)
            continue
        d = OracleData(om)
        d.qpos[:] = q
        d.forward()
        if d.ncon != 1:
            continue
        # world poses of the two geoms
        def pose(name):
            g = gname.index(name)
            if name == "slab":
                return SLAB_POS, np.eye(3)
            ib = BODIES.index(name)
            return q[7 * ib:7 * ib + 3], quat_to_mat(q[7 * ib + 3:7 * ib + 7])
        (pa, Ra), (pb, Rb) = pose(a), pose(b)
        A = _solid_points(a, 6000, rng) @ Ra.T + pa
        Bp = _solid_points(b, 6000, rng) @ Rb.T + pb
        # brute-force distance between the sampled surfaces (valid when the solids do not overlap)
        from scipy.spatial import cKDTree
        dd, _ = cKDTree(Bp).query(A)
        brute = float(dd.min())
        o_dist = float(np.array(d.efc_pos)[d.nefc - 4])      # efc_pos of a contact row = the contact distance
        if o_dist <= 0.002:                                  # overlapping solids: the sampled surfaces say nothing about depth
            continue
        assert abs(o_dist - brute) < 2.5e-3, (pair, o_dist, brute)
        checked += 1
    assert checked >= 20, checked


def raft_model(n_rafts, per_raft):
    """`n_rafts` free bodies, each a flat raft of `per_raft` spheres, resting on the ground plane: n_rafts * per_raft contacts"""
    B = _Builder()
    B.add_geom("plane", 0, 0, (0, 0, 0), collide=1)
    for r in range(n_rafts):
        b = B.add_body("raft%d" % r, 0, (0.5 * r, 0.0, 0.0195), mass=0.2, inertia=(2e-4, 3e-4, 4e-4))
        B.add_joint("raft%d_free" % r, b, 0)
        for k in range(per_raft):
            B.add_geom("s%d_%d" % (r, k), b, SPH, (0.02,), (0.05 * (k % 4) - 0.075, 0.05 * (k // 4) - 0.05, 0.0), collide=1)
    m = B.finish()
    for g in range(len(m.names["geom"])):
        m.arrays["geom_contype"][g], m.arrays["geom_conaffinity"][g] = (1, 0) if g == 0 else (2, 1)
    m.arrays["geom_margin"][:] = 0.002
    set_const(m)
    return m


def case_contact_overflow(lib, dtype):
    """contacts beyond the scratch's capacity are dropped (MuJoCo drops beyond nconmax with a warning) — never silently:
    myo_batch_health counts the substeps it happened in, the states stay finite, and a model inside the capacity counts nothing."""
    mem = Mem(lib)
    cap = 22 if dtype == native.MYO_F64 else 24            # MYO_NREC_F64 / MYO_NCON_MAX record slots (sphere-plane pairs: the base scratch; no limit rows here)
    for n_rafts, per_raft, over in ((1, 12, False), (3, 12, True)):
        assert (n_rafts * per_raft > cap) == over
        cm = compile_model(raft_model(n_rafts, per_raft))
        b = native.Batch(native.Model(cm, lib), None, 2, 0, 0, dtype)
        b.physics_step(None, 5)
        qp = mem.zeros((2, cm.size("nq")))
        b.get_state(qp)
        h = b.health()
        assert np.isfinite(mem.host(qp)).all()
        assert h["protocol_errors"] == 0 and (h["contact_overflows"] > 0) == over, (n_rafts, per_raft, h)
        if not over:                                       # every sphere of the raft is in contact: the count is what the model says
            get, b2 = forward_dump(lib, mem, cm, cm_qpos0(cm), np.zeros(cm.size("nv")), np.zeros(0), np.zeros(0), dtype)
            assert int(get("counts", 4)[0]) == n_rafts * per_raft
            b2.close()
        b.close()


def many_contacts_model():
    """Three rafts of 13 spheres each on the ground plane (39 contacts) and a cylinder lying beside them (an EXTENDED pair: the model gets
    the 48-slot scratch): more contact slots than the 34 the big scratch held until round 5, fewer than the 48 it holds now."""
    B = _Builder()
    B.add_geom("plane", 0, 0, (0, 0, 0), collide=1)
    for r in range(3):
        b = B.add_body("raft%d" % r, 0, (0.5 * r, 0.0, 0.0195), mass=0.2 + 0.05 * r, inertia=(2e-4, 3e-4, 4e-4))
        B.add_joint("raft%d_free" % r, b, 0)
        for k in range(13):
            B.add_geom("s%d_%d" % (r, k), b, SPH, (0.02,), (0.05 * (k % 4) - 0.075, 0.05 * (k // 4) - 0.075, 0.0), collide=1)
    b = B.add_body("log", 0, (2.0, 0.0, 0.0295), mass=0.3, inertia=(3e-4, 3e-4, 2e-4))
    B.add_joint("log_free", b, 0)
    c, sn = np.cos(np.pi / 4), np.sin(np.pi / 4)
    B.add_geom("log_geom", b, CYL, (0.03, 0.08), (0, 0, 0), quat=(c, 0, sn, 0), collide=1)          # axis along x: lying on its side
    m = B.finish()
    for g in range(len(m.names["geom"])):
        m.arrays["geom_contype"][g], m.arrays["geom_conaffinity"][g] = (1, 0) if g == 0 else (2, 1)
    m.arrays["geom_margin"][:] = 0.002
    set_const(m)
    return m


def case_many_contacts(lib, dtype, tol):
    """a state with MORE than 34 contact slots steps like the oracle (which holds 64) — no slot is dropped, the counters
    stay at zero, qacc of the first forward pass and the state after 40 substeps agree."""
    mem = Mem(lib)
    m = many_contacts_model()
    cm = compile_model(m)
    om = OracleModel(cm.to_blob())
    q0 = cm_qpos0(cm)
    d = OracleData(om)
    d.qpos[:] = q0
    d.forward()
    assert 34 < d.ncon <= 48, d.ncon
    get, b2 = forward_dump(lib, mem, cm, q0, np.zeros(cm.size("nv")), np.zeros(0), np.zeros(0), dtype)
    cnt = get("counts", 4)
    assert (int(cnt[0]), int(cnt[1])) == (d.ncon, d.nefc), (cnt, d.ncon, d.nefc)
    qa = np.array(d.qacc)
    assert rel_err(get("qacc", qa.size), qa) < tol, rel_err(get("qacc", qa.size), qa)
    assert b2.health()["contact_overflows"] == 0
    b2.close()
    n = 2
    b = native.Batch(native.Model(cm, lib), None, n, 0, 0, dtype)
    assert b.lds_bytes <= 20480                                       # (the 48-slot scratch: records in the big workspace, eight workgroups per CU)
    b.set_state(mem.arr(np.tile(q0, (n, 1))), mem.zeros((n, om.nv)), mem.zeros((n, 0)), mem.zeros(n))
    qp, qv = mem.zeros((n, om.nq)), mem.zeros((n, om.nv))
    most = 0
    for i in range(40):
        d.step()
        most = max(most, d.ncon)
        b.physics_step(None, 1)
    b.get_state(qp, qv)
    assert rel_err(mem.host(qp)[1], d.qpos) < tol and np.abs(mem.host(qv)[1] - d.qvel).max() < tol * max(1.0, np.abs(d.qvel).max())
    assert most > 34 and b.health() == {"protocol_errors": 0, "contact_overflows": 0, "limit_row_overflows": 0, "contact_slots_wanted": 0}, (most, b.health())
    b.close()


def test_more_than_34_contact_slots_on_emulation(emu_lib):
    case_many_contacts(emu_lib, native.MYO_F64, 1e-9)
    case_many_contacts(emu_lib, native.MYO_MIXED, 1e-4)


@pytest.mark.gpu
def test_more_than_34_contact_slots_on_gpu(hip_lib):
    case_many_contacts(hip_lib, native.MYO_F64, 1e-9)
    case_many_contacts(hip_lib, native.MYO_MIXED, 1e-4)


def cm_qpos0(cm):
    return np.asarray(cm.fields["qpos0"], dtype=np.float64).copy()


def test_contact_overflow_is_counted_on_emulation(emu_lib):
    case_contact_overflow(emu_lib, native.MYO_F64)
    case_contact_overflow(emu_lib, native.MYO_MIXED)


@pytest.mark.gpu
def test_contact_overflow_is_counted_on_gpu(hip_lib):
    case_contact_overflow(hip_lib, native.MYO_F64)
    case_contact_overflow(hip_lib, native.MYO_MIXED)


def crossed_boxes_model(margin=0.002):
    """A static box lying on an edge-up diagonal (ridge along x) and a free box above it whose lowest feature is an edge along y:
    the only contact two such boxes can make is EDGE-EDGE — no vertex of either box is near a face of the other."""
    B = _Builder()
    c, s = np.cos(np.pi / 8), np.sin(np.pi / 8)                        # quaternion of a 45 degree turn about x
    B.add_geom("ridge", 0, BOX, (0.05, 0.03, 0.03), (0.0, 0.0, 0.5), quat=(c, s, 0, 0), collide=1)
    b = B.add_body("upper", 0, (0.0, 0.0, 0.6), mass=0.05, inertia=(2e-5, 3e-5, 4e-5))
    B.add_joint("upper_free", b, 0)
    B.add_geom("vee", b, BOX, (0.05, 0.03, 0.03), collide=1)
    m = B.finish()
    for g in range(2):
        m.arrays["geom_contype"][g], m.arrays["geom_conaffinity"][g] = 1, 1
    m.arrays["geom_margin"][:] = margin
    set_const(m)
    return m


def _quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def case_edge_edge(lib, dtype, tol):
    """Two boxes crossing edge-on: exactly one contact, at the crossing point, normal along +z (geom 1 -> geom 2), distance = the gap
    between the two edges — from the oracle's closed form, from the stepper through the C ABI, and against the analytic values."""
    mem = Mem(lib)
    m = crossed_boxes_model()
    cm = compile_model(m)
    assert cm.dropped_pairs == []
    om = OracleModel(cm.to_blob())
    h = 0.03 * np.sqrt(2.0)                                # a 45-degree box's edge sits this far from its centre
    c, s = np.cos(np.pi / 8), np.sin(np.pi / 8)
    qz = np.array([np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])      # 90 degrees about z: the upper box's long axis along y
    quat = _quat_mul(qz, np.array([c, s, 0, 0]))
    for gap, expect in ((-0.003, 1), (0.001, 1), (0.004, 0)):
        q = np.concatenate([[0.01, -0.02, 0.5 + 2 * h + gap], quat])
        d = OracleData(om)
        d.qpos[:] = q
        d.forward()
        assert d.ncon == expect, (gap, d.ncon)
        get, b = forward_dump(lib, mem, cm, q, np.zeros(6), np.zeros(0), np.zeros(0), dtype)
        cnt = get("counts", 4)
        assert (int(cnt[0]), int(cnt[1])) == (d.ncon, d.nefc), (gap, cnt)
        if expect:
            assert abs(float(np.array(d.efc_pos)[d.nefc - 4]) - gap) < 1e-12, (gap, np.array(d.efc_pos)[:d.nefc])
            assert rel_err(get("efc_aref", d.nefc), np.array(d.efc_aref)[:d.nefc]) < tol
            assert rel_err(get("qacc", om.nv), d.qacc) < tol
            # a downward push on the upper box is answered along +z only (the contact normal), at the crossing point (no torque about it
            # beyond the offset of the box's centre from the crossing point)
            assert np.array(d.qacc)[2] > -9.81 + 1e-3 or gap > 0
        b.close()


def test_edge_edge_box_contact_on_emulation(emu_lib):
    case_edge_edge(emu_lib, native.MYO_F64, 1e-9)
    case_edge_edge(emu_lib, native.MYO_MIXED, 1e-4)


@pytest.mark.gpu
def test_edge_edge_box_contact_on_gpu(hip_lib):
    case_edge_edge(hip_lib, native.MYO_F64, 1e-9)
    case_edge_edge(hip_lib, native.MYO_MIXED, 1e-4)


# ---------------------------------------------------------------------------------------------------------------- condim 1 / 4 / 6
def condim_model(condims=(1, 3, 4, 6), friction=(0.8, 0.02, 0.004)):
    """one free sphere per contact dimension over a plane (pyramidal rows 2 (condim - 1) per contact; condim 1: the
    normal row alone).  The plane has condim 1 and priority 0, the spheres priority 1: the contact takes the sphere's condim."""
    B = _Builder()
    B.add_geom("plane", 0, 0, (0, 0, 0), collide=1, friction=(1.0, 0.005, 0.0001))
    for k, cd in enumerate(condims):
        b = B.add_body("ball%d" % cd, 0, (0.3 * k, 0.0, 0.05), mass=0.1, inertia=(1e-4, 1e-4, 1e-4))
        B.add_joint("ball%d_free" % cd, b, 0, damping=0.0)
        B.add_geom("ball%d" % cd, b, SPH, (0.05,), collide=1, friction=friction)
    m = B.finish()
    for g in range(len(m.names["geom"])):
        m.arrays["geom_contype"][g], m.arrays["geom_conaffinity"][g] = (1, 0) if g == 0 else (2, 1)
        m.arrays["geom_condim"][g] = 1 if g == 0 else condims[g - 1]
        m.arrays["geom_priority"][g] = 0 if g == 0 else 1
    m.arrays["geom_margin"][:] = 0.001
    m.arrays["dof_damping"][:] = 0.0
    set_const(m)
    return m


def case_condim(lib, dtype, tol):
    """contacts of condim 1, 3, 4 and 6 side by side: row counts 1 / 4 / 6 / 10, reference accelerations, regularisers and the solved
    accelerations against the oracle; then 150 steps of spheres that slide, spin and roll — torsional friction must stop the spin of the
    condim-4 / 6 spheres and not of the condim-1 / 3 ones, rolling friction the roll of the condim-6 sphere only."""
    mem = Mem(lib)
    m = condim_model()
    cm = compile_model(m)
    om = OracleModel(cm.to_blob())
    rng = np.random.RandomState(4)
    q = m.qpos0.copy()
    for k in range(4):
        q[7 * k + 2] = 0.05 - 0.0004 * (k + 1)              # resting depths inside the margin / penetrating
    for trial in range(4):
        v = rng.normal(0, 0.5, om.nv) * (trial > 0)
        d = OracleData(om)
        d.qpos[:], d.qvel[:] = q, v
        d.forward()
        assert (d.ncon, d.nefc) == (4, 1 + 4 + 6 + 10)
        get, b = forward_dump(lib, mem, cm, q, v, np.zeros(0), np.zeros(0), dtype)
        cnt = get("counts", 4)
        assert (int(cnt[0]), int(cnt[1])) == (d.ncon, d.nefc), cnt
        assert rel_err(get("efc_aref", d.nefc), np.array(d.efc_aref)[:d.nefc]) < tol
        assert rel_err(get("efc_D", d.nefc), np.array(d.efc_D)[:d.nefc]) < tol
        assert rel_err(get("qacc", om.nv), d.qacc) < tol, (trial, rel_err(get("qacc", om.nv), d.qacc))
        b.close()
    # spin about the vertical and roll along x, then let the contacts act
    v = np.zeros(om.nv)
    for k in range(4):
        v[6 * k + 5] = 20.0                                  # spin about z
        v[6 * k + 0], v[6 * k + 4] = 0.2, 0.2 / 0.05         # rolling without slipping along x
    d = OracleData(om)
    d.qpos[:], d.qvel[:] = q, v
    n = 2
    b = native.Batch(native.Model(cm, lib), None, n, 0, 0, dtype)
    b.set_state(mem.arr(np.tile(q, (n, 1))), mem.arr(np.tile(v, (n, 1))), mem.zeros((n, 0)), mem.zeros(n))
    for _ in range(150):
        d.step()
    b.physics_step(None, 150)
    qv = mem.zeros((n, om.nv)); qp = mem.zeros((n, om.nq))
    b.get_state(qp, qv)
    hv = mem.host(qv)[1]
    ov = np.array(d.qvel)
    assert np.abs(hv - ov).max() <= tol * max(1.0, np.abs(ov).max()) * 50, np.abs(hv - ov).max()
    spin = [abs(ov[6 * k + 5]) for k in range(4)]
    assert spin[0] > 19.9 and spin[1] > 19.9 and spin[2] < 12.0 and spin[3] < 12.0, spin          # torsional friction exists from condim 4 on
    roll = [abs(ov[6 * k + 4]) for k in range(4)]
    assert roll[3] < 0.75 * roll[2], roll                                                         # rolling friction from condim 6 on
    assert b.health() == {"protocol_errors": 0, "contact_overflows": 0, "limit_row_overflows": 0, "contact_slots_wanted": 0}
    b.close()


def test_condim_1_4_6_on_emulation(emu_lib):
    case_condim(emu_lib, native.MYO_F64, 1e-9)
    case_condim(emu_lib, native.MYO_MIXED, 1e-4)


@pytest.mark.gpu
def test_condim_1_4_6_on_gpu(hip_lib):
    case_condim(hip_lib, native.MYO_F64, 1e-9)
    case_condim(hip_lib, native.MYO_MIXED, 1e-4)


# ---------------------------------------------------------------------------------------------------------------- friction loss
def slider_model(frictionloss, mass=0.5):
    """one body on a vertical slide joint under gravity, with dry friction (mj_instantiateFriction: a row whose force saturates at
    +- frictionloss)"""
    B = _Builder()
    b = B.add_body("mass", 0, (0.0, 0.0, 1.0), mass=mass, inertia=(1e-3, 1e-3, 1e-3))
    B.add_joint("slide", b, 2, axis=(0, 0, 1), damping=0.0, armature=0.0)
    B.add_geom("g", b, SPH, (0.02,))
    m = B.finish()
    m.arrays["dof_frictionloss"] = np.array([frictionloss])
    m.arrays["dof_damping"][:] = 0.0
    set_const(m)
    return m


def case_friction_loss(lib, dtype, tol):
    """(i) closed forms on a slider under gravity: friction below the weight -> the row saturates, qacc = -(m g - f) / m exactly; above
    -> it holds, qacc = -m g / (m + D) with D = 1/R of the row (soft constraint).  (ii) the finger model with friction loss on two
    joints and one tendon: stepper vs oracle over a trajectory (rows of both kinds, all three zones of the Huber cost)."""
    mem = Mem(lib)
    mass, g = 0.5, 9.81
    for f in (2.0, 20.0):
        m = slider_model(f, mass)
        cm = compile_model(m)
        om = OracleModel(cm.to_blob())
        d = OracleData(om)
        d.forward()
        assert d.nefc == 1
        imp = 0.9                                                    # impedance at pos - margin = 0: solimp[0]
        D = 1.0 / ((1 - imp) / imp * (1.0 / mass))                   # R = (1 - d) / d * dof_invweight0, invweight0 = 1 / m for a slider
        want = -(mass * g - f) / mass if f < mass * g else -mass * g / (mass + D)
        assert abs(float(d.qacc[0]) - want) <= 1e-9 * abs(want), (f, float(d.qacc[0]), want)
        get, b = forward_dump(lib, mem, cm, np.zeros(1), np.zeros(1), np.zeros(0), np.zeros(0), dtype)
        assert abs(float(get("qacc", 1)[0]) - want) <= tol * abs(want)
        b.close()
    # (ii)
    from myochallenge_amd.mjb import load_mjb
    import os
    mj = load_mjb(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "myo_finger_v0.mjb"))
    mj.arrays["dof_frictionloss"] = np.array([0.02, 0.0, 0.05, 0.0])
    mj.arrays["tendon_frictionloss"] = np.array([0.0, 0.5, 0.0, 0.0, 0.0])
    cm = compile_model(mj)
    om = OracleModel(cm.to_blob())
    d = OracleData(om)
    n = 2
    b = native.Batch(native.Model(cm, lib), None, n, 0, 0, dtype)
    q0 = np.array([0.3, 0.6, 0.6, 0.5])
    d.qpos[:] = q0
    b.set_state(mem.arr(np.tile(q0, (n, 1))), mem.zeros((n, om.nv)), mem.zeros((n, om.na)), mem.zeros(n))
    rng = np.random.RandomState(2)
    qp, qv = mem.zeros((n, om.nq)), mem.zeros((n, om.nv))
    for i in range(150):
        if i % 10 == 0:
            c = rng.uniform(0, 1, om.nu)
        d.ctrl[:] = c
        d.step()
        b.physics_step(mem.arr(np.tile(c, (n, 1))), 1)
    assert d.nefc >= 3
    b.get_state(qp, qv)
    assert rel_err(mem.host(qp)[1], d.qpos) < tol * 10 and np.abs(mem.host(qv)[1] - np.array(d.qvel)).max() < tol * 10 * max(1.0, np.abs(np.array(d.qvel)).max())
    b.close()


def test_friction_loss_on_emulation(emu_lib):
    case_friction_loss(emu_lib, native.MYO_F64, 1e-9)
    case_friction_loss(emu_lib, native.MYO_MIXED, 1e-4)


@pytest.mark.gpu
def test_friction_loss_on_gpu(hip_lib):
    case_friction_loss(hip_lib, native.MYO_F64, 1e-9)
    case_friction_loss(hip_lib, native.MYO_MIXED, 1e-4)


# ---------------------------------------------------------------------------------------------------------------- explicit <pair>s
def case_explicit_pair(lib, dtype, tol):
    """An explicit <contact><pair> replaces the dynamic pair of its two geoms and brings its own margin / gap / solref / solimp / friction /
    condim (mj_contactParam).  A condim-4 pair between the plane and the condim-3 sphere of `condim_model`: the contact appears at the
    PAIR's margin, has 6 rows, its reference acceleration and regulariser follow the pair's solref / solimp / friction — closed forms as in
    tests/test_oracle_closed_forms.py — and the stepper agrees with the oracle."""
    mem = Mem(lib)
    m = condim_model(condims=(3,), friction=(0.8, 0.02, 0.004))
    solref, solimp, margin, fr = (0.01, 1.0), (0.8, 0.9, 0.002, 0.5, 2.0), 0.004, (0.6, 0.03, 0.001)
    m.sizes["npair"] = 1
    m.arrays.update(pair_dim=np.full(1, 4, np.int32), pair_geom1=np.array([0], np.int32), pair_geom2=np.array([1], np.int32),
                    pair_signature=np.zeros(1, np.int32), pair_solref=np.array([solref]), pair_solimp=np.array([solimp]),
                    pair_margin=np.array([margin]), pair_gap=np.zeros(1), pair_friction=np.array([[fr[0], fr[0], fr[1], fr[2], fr[2]]]))
    cm = compile_model(m)
    assert list(cm.x_pair_explicit) == [0]
    om = OracleModel(cm.to_blob())
    mass = 0.1
    for gap_to_plane in (0.003, -0.0005):                    # inside the pair's margin (the geoms' is 0.001) / penetrating
        q = np.array([0, 0, 0.05 + gap_to_plane, 1, 0, 0, 0.0])
        v = np.array([0.1, 0, -0.2, 0, 0, 3.0])
        d = OracleData(om)
        d.qpos[:], d.qvel[:] = q, v
        d.forward()
        assert (d.ncon, d.nefc) == (1, 6), (gap_to_plane, d.ncon, d.nefc)
        r = gap_to_plane - margin                            # pos - margin
        x = min(1.0, abs(r) / solimp[2])
        y = x ** 2 / 0.5 if x <= 0.5 else 1 - (1 - x) ** 2 / 0.5
        imp = solimp[0] + y * (solimp[1] - solimp[0])
        tc = max(solref[0], 2 * 0.002)                       # refsafe: time constant >= 2 timesteps
        k, b = 1 / (solimp[1] ** 2 * tc ** 2), 2 / (solimp[1] * tc)
        R0 = (1 - imp) / imp * (1 / mass) * (1 + fr[0] ** 2)
        D = 1 / (2 * fr[0] ** 2 * R0)
        assert np.allclose(np.array(d.efc_D)[:6], D, rtol=1e-12)
        aref = np.array(d.efc_aref)[:6]
        # rows: n +- mu t1, n +- mu t2, n +- mu_torsion (spin about n); J v: normal -0.2, tangential components of (0.1, 0, 0), spin 3
        assert abs(0.5 * (aref[4] + aref[5]) - (-b * (-0.2) - k * imp * r)) <= 1e-9 * abs(aref[4])
        assert abs(abs(aref[4] - aref[5]) - 2 * b * fr[1] * 3.0) <= 1e-9 * abs(aref[4])
        get, bt = forward_dump(lib, mem, cm, q, v, np.zeros(0), np.zeros(0), dtype)
        cnt = get("counts", 4)
        assert (int(cnt[0]), int(cnt[1])) == (1, 6)
        assert rel_err(get("efc_aref", 6), aref) < tol and rel_err(get("efc_D", 6), np.array(d.efc_D)[:6]) < tol
        assert rel_err(get("qacc", om.nv), d.qacc) < tol
        bt.close()


def test_explicit_pair_on_emulation(emu_lib):
    case_explicit_pair(emu_lib, native.MYO_F64, 1e-9)
    case_explicit_pair(emu_lib, native.MYO_MIXED, 1e-4)


@pytest.mark.gpu
def test_explicit_pair_on_gpu(hip_lib):
    case_explicit_pair(hip_lib, native.MYO_F64, 1e-9)
    case_explicit_pair(hip_lib, native.MYO_MIXED, 1e-4)
