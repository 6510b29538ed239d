"""Host-side model constants (what MuJoCo's compiler pass ``mj_setConst`` fills in).

MuJoCo stores quantities evaluated at ``qpos0`` inside every compiled model
(``tendon_length0``, ``dof_invweight0``, ``body_invweight0``, ``tendon_invweight0``,
``actuator_acc0``, ``stat.meaninertia``).  Models decoded from a
``.mjb`` already carry them.  The synthetic MyoHand stand-in (synth_hand.py) is authored in
Python and needs them computed here: a small fp64 numpy forward-kinematics + spatial-tendon
routine and ``M = sum_b m Jv'Jv + Jw' I Jw + armature``.  Run once at asset-build time, never
in the stepping path.  On the shipped finger model it reproduces the MuJoCo-written values
(tests/test_setconst.py).
"""
from __future__ import annotations

import numpy as np

from .mathutil import axis_angle_quat, quat_mul, quat_to_mat

MINVAL = 1e-15
JNT_FREE, JNT_SLIDE, JNT_HINGE = 0, 2, 3
WRAP_PULLEY, WRAP_SITE, WRAP_SPHERE, WRAP_CYLINDER = 2, 3, 4, 5


def kinematics(m, qpos):
    nb = m.nbody
    xpos = np.zeros((nb, 3))
    xquat = np.zeros((nb, 4))
    xquat[0, 0] = 1
    xmat = np.zeros((nb, 3, 3))
    xmat[0] = np.eye(3)
    xanchor = np.zeros((m.njnt, 3))
    xaxis = np.zeros((m.njnt, 3))
    for b in range(1, nb):
        par = m.body_parentid[b]
        jn, ja = int(m.body_jntnum[b]), int(m.body_jntadr[b])
        if jn == 1 and m.jnt_type[ja] == JNT_FREE:
            qa = m.jnt_qposadr[ja]
            p = qpos[qa:qa + 3].copy()
            q = qpos[qa + 3:qa + 7] / np.linalg.norm(qpos[qa + 3:qa + 7])
            xanchor[ja] = p
            xaxis[ja] = [0, 0, 1]
        else:
            p = xpos[par] + xmat[par] @ m.body_pos[b]
            q = quat_mul(xquat[par], m.body_quat[b])
            for j in range(ja, ja + jn):
                R = quat_to_mat(q)
                anchor = R @ m.jnt_pos[j] + p
                axis = R @ m.jnt_axis[j]
                xanchor[j], xaxis[j] = anchor, axis
                ang = qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]
                if m.jnt_type[j] == JNT_SLIDE:
                    p = p + axis * ang
                else:
                    q = quat_mul(q, axis_angle_quat(m.jnt_axis[j], ang))
                    p = anchor - quat_to_mat(q) @ m.jnt_pos[j]
        q = q / np.linalg.norm(q)
        xpos[b], xquat[b], xmat[b] = p, q, quat_to_mat(q)
    return xpos, xquat, xmat, xanchor, xaxis


def _is_intersect(p1, p2, p3, p4):
    det = (p4[1] - p3[1]) * (p2[0] - p1[0]) - (p4[0] - p3[0]) * (p2[1] - p1[1])
    if abs(det) < MINVAL:
        return False
    a = ((p4[0] - p3[0]) * (p1[1] - p3[1]) - (p4[1] - p3[1]) * (p1[0] - p3[0])) / det
    b = ((p2[0] - p1[0]) * (p1[1] - p3[1]) - (p2[1] - p1[1]) * (p1[0] - p3[0])) / det
    return 0 <= a <= 1 and 0 <= b <= 1


def _wrap_circle(d0, d1, sd, rad):
    sq0, sq1, sqr = d0 @ d0, d1 @ d1, rad * rad
    dif = d1 - d0
    dd = dif @ dif
    if sq0 < sqr or sq1 < sqr or rad < MINVAL or dd < MINVAL:
        return -1.0, None
    a = min(1.0, max(0.0, -(dif @ d0) / dd))
    tmp = a * dif + d0
    if tmp @ tmp > sqr and (sd is None or sd @ tmp >= 0):
        return -1.0, None
    s0, s1 = np.sqrt(sq0 - sqr), np.sqrt(sq1 - sqr)
    sols, good = [], []
    for sgn in (1.0, -1.0):
        a0 = np.array([(d0[0] * sqr + sgn * rad * d0[1] * s0) / sq0,
                       (d0[1] * sqr - sgn * rad * d0[0] * s0) / sq0])
        a1 = np.array([(d1[0] * sqr - sgn * rad * d1[1] * s1) / sq1,
                       (d1[1] * sqr + sgn * rad * d1[0] * s1) / sq1])
        if sd is not None:
            t = a0 + a1
            n = np.linalg.norm(t)
            t = t / n if n >= MINVAL else np.array([1.0, 0.0])
            g = t @ sd
        else:
            g = -((a0 - a1) @ (a0 - a1))
        if _is_intersect(d0, a0, d1, a1):
            g = -10000.0
        sols.append((a0, a1))
        good.append(g)
    i = 0 if good[0] > good[1] else 1
    a0, a1 = sols[i]
    if _is_intersect(d0, a0, d1, a1):
        return -1.0, None
    c = min(1.0, max(-1.0, (a0 @ a1) / sqr))
    return rad * np.arccos(c), (a0, a1)


def wrap_geom(x0, x1, gpos, gmat, radius, wtype, side):
    p0, p1 = gmat.T @ (x0 - gpos), gmat.T @ (x1 - gpos)
    if np.linalg.norm(p0) < MINVAL or np.linalg.norm(p1) < MINVAL:
        return -1.0, None
    if wtype == WRAP_SPHERE:
        ax0 = p0 / np.linalg.norm(p0)
        nrm = np.cross(p0, p1)
        nn = np.linalg.norm(nrm)
        if nn < MINVAL:
            e = np.zeros(3)
            e[int(np.argmin(np.abs(ax0)))] = 1
            nrm = np.cross(ax0, e)
            nrm /= np.linalg.norm(nrm)
        else:
            nrm = nrm / nn
        ax1 = np.cross(nrm, ax0)
        ax1 /= np.linalg.norm(ax1)
    else:
        ax0, ax1 = np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    s0 = np.array([p0 @ ax0, p0 @ ax1])
    s1 = np.array([p1 @ ax0, p1 @ ax1])
    sd = None
    if side is not None:
        ps = gmat.T @ (side - gpos)
        sd = np.array([ps @ ax0, ps @ ax1])
        n = np.linalg.norm(sd)
        sd = (sd / n if n >= MINVAL else np.array([1.0, 0.0])) * radius
    wlen, pts = _wrap_circle(s0, s1, sd, radius)
    if wlen < 0:
        return -1.0, None
    r0 = ax0 * pts[0][0] + ax1 * pts[0][1]
    r1 = ax0 * pts[1][0] + ax1 * pts[1][1]
    if wtype == WRAP_CYLINDER:
        L0, L1 = np.linalg.norm(s0 - pts[0]), np.linalg.norm(s1 - pts[1])
        r0[2] = p0[2] + (p1[2] - p0[2]) * L0 / (L0 + wlen + L1)
        r1[2] = p0[2] + (p1[2] - p0[2]) * (L0 + wlen) / (L0 + wlen + L1)
        wlen = np.sqrt(wlen * wlen + (r1[2] - r0[2]) ** 2)
    return wlen, (gmat @ r0 + gpos, gmat @ r1 + gpos)


def tendon_lengths(m, qpos):
    xpos, xquat, xmat, _, _ = kinematics(m, qpos)
    site_x = xpos[m.site_bodyid] + np.einsum("sij,sj->si", xmat[m.site_bodyid], m.site_pos)
    out = np.zeros(m.ntendon)
    for t in range(m.ntendon):
        adr, num = int(m.tendon_adr[t]), int(m.tendon_num[t])
        length, divisor, j = 0.0, 1.0, 0
        while j < num - 1:
            t0, t1 = int(m.wrap_type[adr + j]), int(m.wrap_type[adr + j + 1])
            id0, id1 = int(m.wrap_objid[adr + j]), int(m.wrap_objid[adr + j + 1])
            if t0 == WRAP_PULLEY or t1 == WRAP_PULLEY:
                if t0 == WRAP_PULLEY:
                    divisor = m.wrap_prm[adr + j]
                j += 1
                continue
            wlen, pts, idg = -1.0, None, -1
            if t1 in (WRAP_SPHERE, WRAP_CYLINDER):
                idg = id1
                id1 = int(m.wrap_objid[adr + j + 2])
                sid = int(round(m.wrap_prm[adr + j + 1]))
                gb = m.geom_bodyid[idg]
                gpos = xpos[gb] + xmat[gb] @ m.geom_pos[idg]
                gmat = xmat[gb] @ quat_to_mat(m.geom_quat[idg])
                wlen, pts = wrap_geom(site_x[id0], site_x[id1], gpos, gmat, m.geom_size[idg, 0], t1,
                                      site_x[sid] if sid >= 0 else None)
            if wlen < 0:
                length += np.linalg.norm(site_x[id1] - site_x[id0]) / divisor
            else:
                length += (np.linalg.norm(pts[0] - site_x[id0]) + wlen +
                           np.linalg.norm(site_x[id1] - pts[1])) / divisor
            j += 2 if idg >= 0 else 1
        out[t] = length
    return out


def _dof_columns(m, xpos, xmat, xanchor, xaxis, body, point):
    """Translational / rotational Jacobian (3 x nv) of a point fixed to ``body``."""
    jp, jr = np.zeros((3, m.nv)), np.zeros((3, m.nv))
    b = body
    while b > 0:
        for j in range(int(m.body_jntadr[b]), int(m.body_jntadr[b] + m.body_jntnum[b])):
            da = int(m.jnt_dofadr[j])
            if m.jnt_type[j] == JNT_FREE:
                jp[:, da:da + 3] = np.eye(3)
                for k in range(3):
                    ax = xmat[b][:, k]
                    jr[:, da + 3 + k] = ax
                    jp[:, da + 3 + k] = np.cross(ax, point - xpos[b])
            elif m.jnt_type[j] == JNT_SLIDE:
                jp[:, da] = xaxis[j]
            else:
                jr[:, da] = xaxis[j]
                jp[:, da] = np.cross(xaxis[j], point - xanchor[j])
        b = int(m.body_parentid[b])
    return jp, jr


def mass_matrix(m, qpos):
    xpos, xquat, xmat, xanchor, xaxis = kinematics(m, qpos)
    M = np.diag(np.asarray(m.dof_armature, float).copy())
    jacs = {}
    for b in range(1, m.nbody):
        xipos = xpos[b] + xmat[b] @ m.body_ipos[b]
        ximat = xmat[b] @ quat_to_mat(m.body_iquat[b])
        jp, jr = _dof_columns(m, xpos, xmat, xanchor, xaxis, b, xipos)
        Iw = ximat @ np.diag(m.body_inertia[b]) @ ximat.T
        M += m.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
        jacs[b] = (jp, jr)
    return M, jacs


def tendon_jacobian(m, qpos, eps=1e-6):
    """Central-difference moment arms (free-joint dofs are not touched by any tendon here)."""
    J = np.zeros((m.ntendon, m.nv))
    for j in range(m.njnt):
        if m.jnt_type[j] == JNT_FREE:
            continue
        qa, da = int(m.jnt_qposadr[j]), int(m.jnt_dofadr[j])
        qp, qm = qpos.copy(), qpos.copy()
        qp[qa] += eps
        qm[qa] -= eps
        J[:, da] = (tendon_lengths(m, qp) - tendon_lengths(m, qm)) / (2 * eps)
    return J


def set_const(m, *, lengthrange_samples=0, seed=0):
    """Fill the qpos0-derived fields of ``m.arrays`` in place (m is an MjbModel-like object)."""
    a = m.arrays
    q0 = np.asarray(m.qpos0, float)
    M, jacs = mass_matrix(m, q0)
    Minv = np.linalg.inv(M)
    a["dof_M0"] = np.diag(M).copy()
    dinv = np.diag(Minv).copy()
    for j in range(m.njnt):
        if m.jnt_type[j] == JNT_FREE:  # MuJoCo averages over the 3 translational / rotational dofs
            da = int(m.jnt_dofadr[j])
            dinv[da:da + 3] = dinv[da:da + 3].mean()
            dinv[da + 3:da + 6] = dinv[da + 3:da + 6].mean()
    a["dof_invweight0"] = dinv
    biw = np.zeros((m.nbody, 2))
    for b, (jp, jr) in jacs.items():
        biw[b] = [np.trace(jp @ Minv @ jp.T) / 3, np.trace(jr @ Minv @ jr.T) / 3]
    a["body_invweight0"] = biw
    if m.ntendon == 0:                     # models without tendons / actuators (the contact test models)
        a["tendon_length0"] = a["tendon_invweight0"] = np.zeros(0)
        a["actuator_acc0"] = a["actuator_length0"] = np.zeros(m.nu)
        m.stat["meaninertia"] = float(np.mean(np.diag(M)))
        return m
    L0 = tendon_lengths(m, q0)
    a["tendon_length0"] = L0
    J = tendon_jacobian(m, q0)
    a["tendon_invweight0"] = np.einsum("ti,ij,tj->t", J, Minv, J)
    mom = np.zeros((m.nu, m.nv))
    for i in range(m.nu):
        mom[i] = m.actuator_gear[i, 0] * J[int(m.actuator_trnid[i, 0])]
    a["actuator_acc0"] = np.linalg.norm(mom @ Minv, axis=1)
    a["actuator_length0"] = m.actuator_gear[:, 0] * L0[m.actuator_trnid[:, 0]]
    m.stat["meaninertia"] = float(np.mean(np.diag(M)))
    if lengthrange_samples:
        rng = np.random.RandomState(seed)
        lo, hi = L0.copy(), L0.copy()
        hinge = [j for j in range(m.njnt) if m.jnt_type[j] != JNT_FREE]
        for _ in range(lengthrange_samples):
            q = q0.copy()
            for j in hinge:
                r = m.jnt_range[j]
                q[int(m.jnt_qposadr[j])] = rng.uniform(r[0], r[1])
            L = tendon_lengths(m, q)
            lo, hi = np.minimum(lo, L), np.maximum(hi, L)
        lr = np.stack([lo, hi], 1)
        a["actuator_lengthrange"] = lr[m.actuator_trnid[:, 0]] * m.actuator_gear[:, :1]
    return m
