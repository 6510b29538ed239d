"""Compile a decoded MuJoCo model into the named-array blob of include/myo_model_blob.h.

The reference hands its model to MuJoCo by file path
(/root/reference/src/envs/__init__.py:17,63); the batched stepper instead takes a flat,
self-describing blob so that the device code never parses files.  Besides copying the MuJoCo
arrays the physics uses, this derives the static tables a wave-per-env stepper wants:

* ``x_body_depth``     tree depth of every body (kinematics runs level by level)
* ``x_pair_geom1/2``   candidate collision pairs after MuJoCo's static filters
                       (contype/conaffinity, same-weld-body, parent-child unless the parent is
                       welded to the world) — what ``mj_collision``'s body-pair pass keeps
* ``x_dof_depth``      number of ancestor dofs (+1) of every dof

Feature gates (raise ``ModelError``): ball joints, equality constraints, friction loss,
elliptic cones, inside-wrapping side sites, mesh/hfield colliders.  Geom pairs whose narrow
phase is not implemented (ellipsoid-ellipsoid, cylinder-cylinder, box-cylinder, box-ellipsoid, anything with a mesh or height field) make ``compile_model`` raise
``UnsupportedContactsError`` unless the caller opts in with ``unsupported_contacts="drop"``; the dropped pairs are
then listed in ``CompiledModel.dropped_pairs``.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

from .mjb import MjbModel

BLOB_MAGIC = 0x4D4F594D
BLOB_VERSION = 1
NAME_LEN = 40

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
WRAP_JOINT, WRAP_PULLEY, WRAP_SITE, WRAP_SPHERE, WRAP_CYLINDER = 1, 2, 3, 4, 5

SUPPORTED_PAIRS = {
    (GEOM_PLANE, GEOM_SPHERE), (GEOM_PLANE, GEOM_CAPSULE), (GEOM_PLANE, GEOM_ELLIPSOID), (GEOM_PLANE, GEOM_CYLINDER),
    (GEOM_SPHERE, GEOM_SPHERE), (GEOM_SPHERE, GEOM_CAPSULE), (GEOM_SPHERE, GEOM_BOX), (GEOM_SPHERE, GEOM_CYLINDER),
    (GEOM_SPHERE, GEOM_ELLIPSOID),
    (GEOM_CAPSULE, GEOM_CAPSULE), (GEOM_CAPSULE, GEOM_BOX), (GEOM_CAPSULE, GEOM_CYLINDER), (GEOM_CAPSULE, GEOM_ELLIPSOID),
    (GEOM_BOX, GEOM_BOX),            # vertex-face contacts: listed as 16 candidates per geom pair (x_pair_sub)
}

_INT_FIELDS = [
    "body_parentid", "body_rootid", "body_weldid", "body_jntnum", "body_jntadr", "body_dofnum",
    "body_dofadr", "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited",
    "dof_bodyid", "dof_jntid", "dof_parentid", "geom_type", "geom_contype", "geom_conaffinity",
    "geom_condim", "geom_bodyid", "geom_priority", "site_bodyid", "tendon_adr", "tendon_num",
    "tendon_limited", "wrap_type", "wrap_objid", "actuator_trntype", "actuator_dyntype",
    "actuator_gaintype", "actuator_biastype", "actuator_trnid", "actuator_ctrllimited",
    "actuator_forcelimited",
]
_F64_FIELDS = [
    "qpos0", "qpos_spring", "body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass",
    "body_inertia", "body_invweight0", "jnt_solref", "jnt_solimp", "jnt_pos", "jnt_axis",
    "jnt_stiffness", "jnt_range", "jnt_margin", "dof_armature", "dof_damping",
    "dof_invweight0", "geom_solmix", "geom_solref", "geom_solimp", "geom_size", "geom_rbound",
    "geom_pos", "geom_quat", "geom_friction", "geom_margin", "geom_gap", "site_pos",
    "tendon_solref_lim", "tendon_solimp_lim", "tendon_range", "tendon_margin",
    "tendon_stiffness", "tendon_damping", "tendon_lengthspring", "tendon_invweight0",
    "wrap_prm", "actuator_dynprm", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange",
    "actuator_forcerange", "actuator_gear", "actuator_acc0", "actuator_lengthrange",
]
_SIZES = ["nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "ntendon", "nwrap"]


class ModelError(ValueError):
    pass


@dataclass
class CompiledModel:
    fields: Dict[str, np.ndarray]
    names: Dict[str, List[str]] = field(default_factory=dict)
    dropped_pairs: List[Tuple[int, int]] = field(default_factory=list)

    def __getattr__(self, key):
        f = self.__dict__.get("fields", {})
        if key in f:
            return f[key]
        raise AttributeError(key)

    def size(self, name: str) -> int:
        return int(self.fields["sizes"][_SIZES.index(name)]) if name in _SIZES else int(self.fields[name][0])

    def name2id(self, kind: str, name: str) -> int:
        try:
            return self.names[kind].index(name)
        except (ValueError, KeyError) as exc:
            raise KeyError(f"no {kind} named {name!r}") from exc

    def to_blob(self) -> bytes:
        items = list(self.fields.items())
        head = 16 + 56 * len(items)
        head = (head + 7) // 8 * 8
        payload = bytearray()
        table = bytearray()
        for name, arr in items:
            if arr.dtype == np.int32:
                dt = 0
            elif arr.dtype == np.float64:
                dt = 1
            else:
                raise ModelError(f"field {name} has dtype {arr.dtype}")
            raw = np.ascontiguousarray(arr).tobytes()
            off = head + len(payload)
            nm = name.encode()
            if len(nm) >= NAME_LEN:
                raise ModelError(f"field name too long: {name}")
            table += struct.pack(f"<{NAME_LEN}sIIQ", nm, dt, arr.size, off)
            payload += raw
            payload += b"\0" * (-len(raw) % 8)
        total = head + len(payload)
        out = struct.pack("<IIII", BLOB_MAGIC, BLOB_VERSION, len(items), total) + bytes(table)
        out += b"\0" * (head - len(out))
        return out + bytes(payload)


def _depths(parent: np.ndarray) -> np.ndarray:
    d = np.zeros(len(parent), np.int32)
    for i in range(1, len(parent)):
        d[i] = d[parent[i]] + 1
    return d


DSBL_FILTERPARENT, DSBL_REFSAFE = 1 << 9, 1 << 11      # mjtDisableBit (MuJoCo 2.1)
ENBL_OVERRIDE = 1 << 0                                  # mjtEnableBit


def collision_pairs(m: MjbModel):
    """Static part of MuJoCo's collision filtering (mj_collision body-pair pass [3P-RECALL]): bodies of one weld group never
    collide; parent-child weld groups neither unless mjDSBL_FILTERPARENT is set; ``<contact><exclude>`` body pairs
    (``exclude_signature`` = ((body1 + 1) << 16) + body2 + 1, body1 < body2) are skipped; contype / conaffinity per geom pair."""
    ng = m.ngeom
    gb = m.geom_bodyid
    weld = m.body_weldid
    par = m.body_parentid
    excluded = set(int(x) for x in np.asarray(m.arrays.get("exclude_signature", [])).reshape(-1))
    filterparent = not (int(m.opt.get("disableflags", 0)) & DSBL_FILTERPARENT)
    pairs, dropped = [], []

    def emit(a, b, xp):
        ta, tb = int(m.geom_type[a]), int(m.geom_type[b])
        if ta > tb:
            a, b, ta, tb = b, a, tb, ta                       # MuJoCo orders a pair by geom type
        if (ta, tb) == (GEOM_PLANE, GEOM_PLANE):
            return
        if (ta, tb) == (GEOM_BOX, GEOM_BOX):
            pairs.extend((a, b, 1 + v, xp) for v in range(17))   # sub = 1 + v: vertex v of a vs b; 9 + v: vertex v of b vs a; 17: the edge-edge candidate
        elif (ta, tb) in SUPPORTED_PAIRS:
            pairs.append((a, b, 0, xp))
        else:
            dropped.append((a, b))

    # explicit <contact><pair> entries (no contype / conaffinity / parent / exclude filtering applies to them): opt.collision 0 = all,
    # 1 = predefined only, 2 = dynamic only.  With "all", mj_collision (MuJoCo 2.1) merges the predefined list into its BODY-pair sweep by
    # pair_signature = ((body1 + 1) << 16) + body2 + 1 — the same key as exclude_signature — and a body pair that has predefined pairs gets
    # ONLY those: every dynamic geom pair between the two bodies is skipped, not just the pair of the same two geoms
    col = int(m.opt.get("collision", 0))
    explicit = set()
    npair = int(m.sizes.get("npair", 0)) if col != 2 else 0
    for k in range(npair):
        a, b = int(m.arrays["pair_geom1"][k]), int(m.arrays["pair_geom2"][k])
        explicit.add((min(int(gb[a]), int(gb[b])), max(int(gb[a]), int(gb[b]))))
        emit(a, b, k)
    if col == 1:
        return pairs, dropped
    for g1 in range(ng):
        for g2 in range(g1 + 1, ng):
            b1, b2 = int(gb[g1]), int(gb[g2])
            if (min(b1, b2), max(b1, b2)) in explicit:
                continue
            w1, w2 = int(weld[b1]), int(weld[b2])
            if w1 == w2:
                continue
            if excluded and (((min(b1, b2) + 1) << 16) + max(b1, b2) + 1) in excluded:
                continue
            wp1, wp2 = int(weld[par[w1]]), int(weld[par[w2]])
            if filterparent and w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                continue
            ct1, ca1 = int(m.geom_contype[g1]), int(m.geom_conaffinity[g1])
            ct2, ca2 = int(m.geom_contype[g2]), int(m.geom_conaffinity[g2])
            if not ((ct1 & ca2) or (ct2 & ca1)):
                continue
            emit(g1, g2, -1)
    return pairs, dropped


SOLVER_PGS, SOLVER_CG, SOLVER_NEWTON = 0, 1, 2
DYN_NONE, DYN_INTEGRATOR, DYN_FILTER, DYN_MUSCLE, DYN_USER = range(5)
TRN_JOINT, TRN_JOINTINPARENT, TRN_SLIDERCRANK, TRN_TENDON, TRN_SITE = range(5)


def unsupported_features(m: MjbModel) -> List[dict]:
    """Everything in `m` that MuJoCo 2.1's mj_step honours and this stepper does not restate, each with a count — the whole list,
    not the first hit (the environments are registered on .mjb files nobody here has seen,
    /root/reference/src/envs/__init__.py:17,63; their user gets ONE report).  Entries: {"key", "count", "message"}.  compile_model
    refuses a model with any entry (colliding pairs without a narrow phase: unless unsupported_contacts="drop");
    csrc/myo_mjb.h:to_blob builds the same list for the C++ route (tests/test_mjb_and_model.py compares the two)."""
    out: List[dict] = []

    def add(key, count, message):
        if count:
            out.append({"key": key, "count": int(count), "message": message})
    o = m.opt
    a = m.arrays
    add("ball_joints", int(np.sum(np.asarray(m.jnt_type) == JNT_BALL)), "ball joints are not supported")
    add("equality", int(m.sizes.get("neq", 0)), "equality constraints are not supported")
    add("cone", int(o.get("cone", 0) != 0), "only pyramidal friction cones are supported (opt.cone = elliptic)")
    solver = int(o.get("solver", SOLVER_NEWTON))
    add("solver", int(solver != SOLVER_NEWTON), f"opt.solver = {('PGS', 'CG', 'Newton')[solver] if 0 <= solver < 3 else solver}: this stepper restates "
        "mj_solNewton only (a model asking for PGS / CG would be stepped with another algorithm)")
    add("noslip", int(int(o.get("noslip_iterations", 0)) > 0), f"opt.noslip_iterations = {int(o.get('noslip_iterations', 0))}: the noslip post-solver is not implemented")
    fluid = [k for k in ("density", "viscosity") if float(o.get(k, 0.0)) != 0.0]
    if any(float(x) != 0.0 for x in np.atleast_1d(o.get("wind", [0.0, 0.0, 0.0]))):
        fluid.append("wind")
    add("fluid", len(fluid), f"opt.{' / '.join(fluid)} non-zero: fluid forces in mj_passive are not implemented")
    add("integrator", int(int(o.get("integrator", 0)) not in (0, 1)), f"opt.integrator = {int(o.get('integrator', 0))}: Euler (0) and RK4 (1) are implemented")
    cd = np.asarray(m.geom_condim)[(np.asarray(m.geom_contype) | np.asarray(m.geom_conaffinity)) != 0]
    add("condim", int(np.sum(~np.isin(cd, (1, 3, 4, 6)))), "contact dimensions (condim) other than 1, 3, 4, 6 do not exist in MuJoCo")
    dis, enb = int(o.get("disableflags", 0)), int(o.get("enableflags", 0))
    add("disableflags", int(bool(dis & ~(DSBL_FILTERPARENT | DSBL_REFSAFE))), f"opt.disableflags = {dis:#x}: only filterparent and refsafe can be disabled in this stepper")
    add("override", int(bool(enb & ENBL_OVERRIDE)), "opt.enableflags: contact override is not supported")
    # explicit <contact><pair> entries
    col = int(o.get("collision", 0))
    npair_x = int(m.sizes.get("npair", 0)) if col != 2 else 0
    short = [name for name, w in (("pair_geom1", 1), ("pair_geom2", 1), ("pair_dim", 1), ("pair_solref", 2), ("pair_solimp", 5), ("pair_margin", 1),
                                  ("pair_gap", 1), ("pair_friction", 5)) if npair_x and np.asarray(a.get(name, [])).size < npair_x * w]
    add("pair_arrays", len(short), f"npair = {npair_x} but {', '.join(short)} hold(s) fewer values")
    if npair_x and not short:
        g1, g2 = np.asarray(a["pair_geom1"][:npair_x]).astype(int), np.asarray(a["pair_geom2"][:npair_x]).astype(int)
        add("pair_geom_range", int(np.sum((g1 < 0) | (g1 >= m.ngeom) | (g2 < 0) | (g2 >= m.ngeom))), "an explicit contact pair names a geom out of range")
        fr = np.asarray(a["pair_friction"], np.float64).reshape(-1, 5)[:npair_x]
        add("pair_anisotropic", int(np.sum((fr[:, 0] != fr[:, 1]) | (fr[:, 3] != fr[:, 4]))), "explicit contact pair(s) with anisotropic friction are not supported")
        add("pair_condim", int(np.sum(~np.isin(np.asarray(a["pair_dim"][:npair_x]).astype(int), (1, 3, 4, 6)))), "explicit contact pair(s) with a condim other than 1, 3, 4, 6")
    # actuators: tendon transmissions; no / muscle activation dynamics; fixed / muscle gain; none / affine / muscle bias
    if int(m.sizes.get("nu", 0)):
        add("transmission", int(np.sum(np.asarray(a["actuator_trntype"]).astype(int) != TRN_TENDON)), "only tendon transmissions are supported")
        add("actuator_dyn", int(np.sum(~np.isin(np.asarray(a["actuator_dyntype"]).astype(int), (DYN_NONE, DYN_MUSCLE)))),
            "actuator dyntype integrator / filter / user is not implemented (none and muscle are)")
        add("actuator_gain", int(np.sum(~np.isin(np.asarray(a["actuator_gaintype"]).astype(int), (0, 1)))), "actuator gaintype user is not implemented (fixed and muscle are)")
        add("actuator_bias", int(np.sum(~np.isin(np.asarray(a["actuator_biastype"]).astype(int), (0, 1, 2)))), "actuator biastype user is not implemented (none, affine and muscle are)")
    # tendons: spatial only; a wrapping side site inside its wrap geom would need MuJoCo's inside-wrap Newton iteration
    n_fixed, n_inside = 0, 0
    for t in range(int(m.sizes.get("ntendon", 0))):
        adr, num = int(m.tendon_adr[t]), int(m.tendon_num[t])
        for w in range(adr, adr + num):
            wt = int(m.wrap_type[w])
            if wt == WRAP_JOINT:
                n_fixed += 1
            if wt in (WRAP_SPHERE, WRAP_CYLINDER) and m.wrap_prm[w] >= 0:
                sid, gid = int(round(m.wrap_prm[w])), int(m.wrap_objid[w])
                if 0 <= sid < m.nsite and 0 <= gid < m.ngeom and m.site_bodyid[sid] == m.geom_bodyid[gid]:
                    d = np.asarray(m.site_pos[sid], float) - np.asarray(m.geom_pos[gid], float)
                    if wt == WRAP_CYLINDER:
                        from .mathutil import quat_to_mat
                        R = quat_to_mat(m.geom_quat[gid])
                        d = R.T @ d
                        d[2] = 0.0
                    if np.linalg.norm(d) < m.geom_size[gid, 0]:
                        n_inside += 1
    add("fixed_tendons", n_fixed, "fixed (joint) tendons are not supported")
    add("side_site_inside", n_inside, "a wrapping side site lies inside its wrap geom (MuJoCo's inside-wrap iteration is not implemented)")
    fl = a.get("dof_frictionloss")
    if fl is not None and int(m.sizes.get("nv", 0)):
        free = np.asarray(m.jnt_type)[np.asarray(m.dof_jntid).astype(int)] == JNT_FREE
        add("frictionloss_free", int(np.sum((np.asarray(fl, float).reshape(-1) > 0) & free)), "friction loss on the dofs of a free joint is not supported")
    if not any(x["key"] in ("pair_arrays", "pair_geom_range") for x in out):
        try:
            _, dropped = collision_pairs(m)
        except Exception:      # noqa: BLE001 — a corrupt model: the loaders report that
            dropped = []
        if dropped:
            kinds = {}
            for ga, gb_ in dropped:
                k = tuple(sorted((int(m.geom_type[ga]), int(m.geom_type[gb_]))))
                kinds[k] = kinds.get(k, 0) + 1
            tn = ("plane", "hfield", "sphere", "capsule", "ellipsoid", "cylinder", "box", "mesh")
            add("contact_pairs", len(dropped), f"{len(dropped)} colliding geom pair(s) have no narrow phase in this stepper ("
                + ", ".join(f"{n} x {tn[k[0]]}-{tn[k[1]]}" for k, n in sorted(kinds.items())) + ")")
    return out


class UnsupportedContactsError(ValueError):
    """The model has colliding geom pairs whose narrow phase the stepper does not implement."""


def compile_model(m: MjbModel, *, integrator: int | None = None, unsupported_contacts: str = "error",
                  allow_other_solver: bool = False) -> CompiledModel:
    """unsupported_contacts: what to do with colliding geom pairs that have no narrow phase here (meshes, height fields,
    cylinder-cylinder, ellipsoid-ellipsoid and the like; the pairs of SUPPORTED_PAIRS and box-box — 16 vertex-face candidates plus
    one edge-edge candidate from a separating-axis test — have one): "error" (default) refuses the model — a contact MuJoCo would generate
    must not vanish silently — "drop" compiles without them and lists them in ``CompiledModel.dropped_pairs``
    (an explicit opt-in: the physics then differs from MuJoCo's whenever such a pair would touch)."""
    if unsupported_contacts not in ("error", "drop"):
        raise ValueError("unsupported_contacts must be 'error' or 'drop'")
    f: Dict[str, np.ndarray] = {}
    f["sizes"] = np.array([m.sizes[k] for k in _SIZES], np.int32)
    # what mj_step would do differently and this stepper does not restate is refused, never ignored — ALL of it at once
    # (unsupported_features: the same list `python -m myochallenge_amd.model --check file.mjb` prints)
    # (allow_other_solver: the caller's explicit choice to step a PGS / CG / noslip model with this stepper's Newton solver — ADVICE r05)
    feats = [x for x in unsupported_features(m) if x["key"] != "contact_pairs" and not (allow_other_solver and x["key"] in ("solver", "noslip"))]
    if feats:
        raise ModelError("; ".join(x["message"] for x in feats))
    col = int(m.opt.get("collision", 0))
    npair_x = int(m.sizes.get("npair", 0)) if col != 2 else 0
    for name in _INT_FIELDS:
        f[name] = np.ascontiguousarray(m.arrays[name]).astype(np.int32).reshape(-1)
    for name in _F64_FIELDS:
        f[name] = np.ascontiguousarray(m.arrays[name]).astype(np.float64).reshape(-1)
    # friction loss (mj_instantiateFriction rows): the coefficients and the rows' solver parameters (MuJoCo's defaults where a
    # model source carries none)
    nv_, nt_ = int(m.sizes["nv"]), int(m.sizes["ntendon"])
    for name, cnt, default in (("dof_frictionloss", nv_, (0.0,)), ("dof_solref", nv_, (0.02, 1.0)), ("dof_solimp", nv_, (0.9, 0.95, 0.001, 0.5, 2.0)),
                               ("tendon_frictionloss", nt_, (0.0,)), ("tendon_solref_fri", nt_, (0.02, 1.0)),
                               ("tendon_solimp_fri", nt_, (0.9, 0.95, 0.001, 0.5, 2.0))):
        a = m.arrays.get(name)
        f[name] = (np.tile(np.array(default, np.float64), cnt) if a is None else np.ascontiguousarray(a).astype(np.float64).reshape(-1))
    # site orientations: carried for the task layers' checks (the die-reorient observation reads site_xmat in MyoSuite; this
    # stepper takes the body's orientation and therefore requires identity site frames, envs/reorient.py:make_reorient_cfg)
    sq = m.arrays.get("site_quat")
    f["site_quat"] = (np.tile([1.0, 0, 0, 0], (m.sizes["nsite"], 1)) if sq is None else np.asarray(sq, np.float64)).reshape(-1)
    f["x_body_depth"] = _depths(m.body_parentid)
    dd = np.zeros(m.nv, np.int32)
    for i in range(m.nv):
        p = int(m.dof_parentid[i])
        dd[i] = 1 if p < 0 else dd[p] + 1
    f["x_dof_depth"] = dd
    pairs, dropped = collision_pairs(m)
    if dropped and unsupported_contacts == "error":
        gname = m.names.get("geom", []) if isinstance(m.names, dict) else []
        nm = lambda g: gname[g] if g < len(gname) and gname[g] else f"geom{g}"
        some = ", ".join(f"{nm(a)}(type {int(m.geom_type[a])})-{nm(b)}(type {int(m.geom_type[b])})" for a, b in dropped[:6])
        raise UnsupportedContactsError(
            f"{len(dropped)} colliding geom pair(s) have no narrow phase in this stepper: {some}"
            f"{' ...' if len(dropped) > 6 else ''}.  Pass unsupported_contacts='drop' to compile without them.")
    std = {(GEOM_PLANE, GEOM_SPHERE), (GEOM_PLANE, GEOM_CAPSULE), (GEOM_SPHERE, GEOM_SPHERE), (GEOM_SPHERE, GEOM_CAPSULE),
           (GEOM_SPHERE, GEOM_BOX), (GEOM_CAPSULE, GEOM_CAPSULE)}
    is_std = lambda pr: (int(m.geom_type[pr[0]]), int(m.geom_type[pr[1]])) in std
    pairs = [pr for pr in pairs if is_std(pr)] + [pr for pr in pairs if not is_std(pr)]      # the stepper runs the primitive pairs first
    pa = np.array(pairs, np.int32).reshape(-1, 4)
    f["x_pair_geom1"] = np.ascontiguousarray(pa[:, 0])
    f["x_pair_geom2"] = np.ascontiguousarray(pa[:, 1])
    f["x_pair_sub"] = np.ascontiguousarray(pa[:, 2])
    # explicit <pair> parameters (mj_contactParam takes them instead of mixing the two geoms'): per pair ROW the index of its explicit
    # entry (-1: a dynamic pair), and the entries: condim, margin, gap, solref, solimp, friction (sliding, torsional, rolling)
    f["x_pair_explicit"] = np.ascontiguousarray(pa[:, 3])
    f["x_xp_dim"] = np.asarray(m.arrays["pair_dim"][:npair_x], np.int32).reshape(-1) if npair_x else np.zeros(0, np.int32)
    xf = (lambda name, w: np.asarray(m.arrays[name], np.float64).reshape(-1, w)[:npair_x].reshape(-1) if npair_x else np.zeros(0))
    f["x_xp_margin"], f["x_xp_gap"] = xf("pair_margin", 1), xf("pair_gap", 1)
    f["x_xp_solref"], f["x_xp_solimp"] = xf("pair_solref", 2), xf("pair_solimp", 5)
    f["x_xp_friction"] = (np.asarray(m.arrays["pair_friction"], np.float64).reshape(-1, 5)[:npair_x][:, [0, 2, 3]].reshape(-1) if npair_x else np.zeros(0))
    o = m.opt
    f["opt_int"] = np.array([integrator if integrator is not None else o["integrator"],
                             o["cone"], o["iterations"], o["disableflags"]], np.int32)
    f["opt_f64"] = np.array([o["timestep"], o["tolerance"], o["impratio"], *o["gravity"],
                             o["o_margin"], m.stat["meaninertia"]], np.float64)
    return CompiledModel(fields=f, names=dict(m.names), dropped_pairs=dropped)


def _main(argv=None) -> int:
    """``python -m myochallenge_amd.model --check file.mjb``: every feature of the model this stepper refuses, with counts, at once."""
    import argparse
    import json
    from .mjb import load_mjb
    ap = argparse.ArgumentParser(prog="python -m myochallenge_amd.model")
    ap.add_argument("--check", metavar="FILE.mjb", required=True, help="list every unsupported feature of the model (exit code 1 if there is one)")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args(argv)
    m = load_mjb(args.check)
    feats = unsupported_features(m)
    if args.json:
        print(json.dumps({"file": args.check, "sizes": {k: int(m.sizes[k]) for k in _SIZES}, "unsupported": feats}))
    else:
        print(f"{args.check}: " + ", ".join(f"{k} {int(m.sizes[k])}" for k in _SIZES))
        if not feats:
            print("  every feature of this model is implemented")
        for x in feats:
            print(f"  [{x['key']}] x{x['count']}: {x['message']}")
    return 1 if feats else 0


if __name__ == "__main__":
    raise SystemExit(_main())
