"""Reader for MuJoCo-2.1 binary models (``.mjb``).

The reference registers its environments on ``.mjb`` files
(/root/reference/src/envs/__init__.py:17,63 ``model_path=…/myo_hand_baoding.mjb``) that
MuJoCo's ``mj_loadModel`` would parse.  MuJoCo is a third-party dependency that is not
available to this project, so this module re-implements the file layout
(SURVEY.md Appendix A.4): a 4-int header, 57 size ints, the ``mjOption`` block, the
``mjVisual``/``mjStatistic`` blocks and then ``nbuffer`` bytes holding every model array in
``MJMODEL_POINTERS`` order with natural alignment.

The layout was validated byte-exactly (computed end offset == ``nbuffer``) on the three
models shipped under /root/reference/data/myosuite/assets/{finger,basic}.  The
camera/light/mesh/skin/hfield/texture/material/pair/exclude/equality/sensor/... groups are
empty in those files; their order below follows the public MuJoCo 2.1 ``mjxmacro.h`` and is
checked only through the same end-offset identity when a user supplies a bigger model.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

MJB_MAGIC = 54321
N_SIZE_INTS = 57
N_POINTERS = 266

SIZE_NAMES = (
    "nq nv nu na nbody njnt ngeom nsite ncam nlight nmesh nmeshvert nmeshtexvert nmeshface "
    "nmeshgraph nskin nskinvert nskintexvert nskinface nskinbone nskinbonevert nhfield "
    "nhfielddata ntex ntexdata nmat npair nexclude neq ntendon nwrap nsensor nnumeric "
    "nnumericdata ntext ntextdata ntuple ntupledata nkey nmocap nuser_body nuser_jnt nuser_geom "
    "nuser_site nuser_cam nuser_tendon nuser_actuator nuser_sensor nnames nM nemax njmax nconmax "
    "nstack nuserdata nsensordata nbuffer"
).split()
assert len(SIZE_NAMES) == N_SIZE_INTS

OPT_DOUBLES = (
    ("timestep", 1), ("apirate", 1), ("impratio", 1), ("tolerance", 1),
    ("noslip_tolerance", 1), ("mpr_tolerance", 1), ("gravity", 3), ("wind", 3),
    ("magnetic", 3), ("density", 1), ("viscosity", 1), ("o_margin", 1),
    ("o_solref", 2), ("o_solimp", 5),
)
OPT_INTS = (
    "integrator collision cone jacobian solver iterations noslip_iterations "
    "mpr_iterations disableflags enableflags"
).split()

# (type, name, rows, cols).  type: d=f64 i=i32 b=u8 f=f32 c=char.  rows/cols are size names,
# integer literals, or simple products "a*b".
_SPEC = """
d qpos0 nq 1
d qpos_spring nq 1
i body_parentid nbody 1
i body_rootid nbody 1
i body_weldid nbody 1
i body_mocapid nbody 1
i body_jntnum nbody 1
i body_jntadr nbody 1
i body_dofnum nbody 1
i body_dofadr nbody 1
i body_geomnum nbody 1
i body_geomadr nbody 1
b body_simple nbody 1
b body_sameframe nbody 1
d body_pos nbody 3
d body_quat nbody 4
d body_ipos nbody 3
d body_iquat nbody 4
d body_mass nbody 1
d body_subtreemass nbody 1
d body_inertia nbody 3
d body_invweight0 nbody 2
d body_user nbody nuser_body
i jnt_type njnt 1
i jnt_qposadr njnt 1
i jnt_dofadr njnt 1
i jnt_bodyid njnt 1
i jnt_group njnt 1
b jnt_limited njnt 1
d jnt_solref njnt 2
d jnt_solimp njnt 5
d jnt_pos njnt 3
d jnt_axis njnt 3
d jnt_stiffness njnt 1
d jnt_range njnt 2
d jnt_margin njnt 1
d jnt_user njnt nuser_jnt
i dof_bodyid nv 1
i dof_jntid nv 1
i dof_parentid nv 1
i dof_Madr nv 1
i dof_simplenum nv 1
d dof_solref nv 2
d dof_solimp nv 5
d dof_frictionloss nv 1
d dof_armature nv 1
d dof_damping nv 1
d dof_invweight0 nv 1
d dof_M0 nv 1
i geom_type ngeom 1
i geom_contype ngeom 1
i geom_conaffinity ngeom 1
i geom_condim ngeom 1
i geom_bodyid ngeom 1
i geom_dataid ngeom 1
i geom_matid ngeom 1
i geom_group ngeom 1
i geom_priority ngeom 1
b geom_sameframe ngeom 1
d geom_solmix ngeom 1
d geom_solref ngeom 2
d geom_solimp ngeom 5
d geom_size ngeom 3
d geom_rbound ngeom 1
d geom_pos ngeom 3
d geom_quat ngeom 4
d geom_friction ngeom 3
d geom_margin ngeom 1
d geom_gap ngeom 1
d geom_user ngeom nuser_geom
f geom_rgba ngeom 4
i site_type nsite 1
i site_bodyid nsite 1
i site_matid nsite 1
i site_group nsite 1
b site_sameframe nsite 1
d site_size nsite 3
d site_pos nsite 3
d site_quat nsite 4
d site_user nsite nuser_site
f site_rgba nsite 4
i cam_mode ncam 1
i cam_bodyid ncam 1
i cam_targetbodyid ncam 1
d cam_pos ncam 3
d cam_quat ncam 4
d cam_poscom0 ncam 3
d cam_pos0 ncam 3
d cam_mat0 ncam 9
d cam_fovy ncam 1
d cam_ipd ncam 1
d cam_user ncam nuser_cam
i light_mode nlight 1
i light_bodyid nlight 1
i light_targetbodyid nlight 1
b light_directional nlight 1
b light_castshadow nlight 1
b light_active nlight 1
d light_pos nlight 3
d light_dir nlight 3
d light_poscom0 nlight 3
d light_pos0 nlight 3
d light_dir0 nlight 3
f light_attenuation nlight 3
f light_cutoff nlight 1
f light_exponent nlight 1
f light_ambient nlight 3
f light_diffuse nlight 3
f light_specular nlight 3
i mesh_vertadr nmesh 1
i mesh_vertnum nmesh 1
i mesh_texcoordadr nmesh 1
i mesh_faceadr nmesh 1
i mesh_facenum nmesh 1
i mesh_graphadr nmesh 1
f mesh_vert nmeshvert 3
f mesh_normal nmeshvert 3
f mesh_texcoord nmeshtexvert 2
i mesh_face nmeshface 3
i mesh_graph nmeshgraph 1
i skin_matid nskin 1
f skin_rgba nskin 4
f skin_inflate nskin 1
i skin_vertadr nskin 1
i skin_vertnum nskin 1
i skin_texcoordadr nskin 1
i skin_faceadr nskin 1
i skin_facenum nskin 1
i skin_boneadr nskin 1
i skin_bonenum nskin 1
f skin_vert nskinvert 3
f skin_texcoord nskintexvert 2
i skin_face nskinface 3
i skin_bonevertadr nskinbone 1
i skin_bonevertnum nskinbone 1
f skin_bonebindpos nskinbone 3
f skin_bonebindquat nskinbone 4
i skin_bonebodyid nskinbone 1
i skin_bonevertid nskinbonevert 1
f skin_bonevertweight nskinbonevert 1
d hfield_size nhfield 4
i hfield_nrow nhfield 1
i hfield_ncol nhfield 1
i hfield_adr nhfield 1
f hfield_data nhfielddata 1
i tex_type ntex 1
i tex_height ntex 1
i tex_width ntex 1
i tex_adr ntex 1
b tex_rgb ntexdata 1
i mat_texid nmat 1
b mat_texuniform nmat 1
f mat_texrepeat nmat 2
f mat_emission nmat 1
f mat_specular nmat 1
f mat_shininess nmat 1
f mat_reflectance nmat 1
f mat_rgba nmat 4
i pair_dim npair 1
i pair_geom1 npair 1
i pair_geom2 npair 1
i pair_signature npair 1
d pair_solref npair 2
d pair_solimp npair 5
d pair_margin npair 1
d pair_gap npair 1
d pair_friction npair 5
i exclude_signature nexclude 1
i eq_type neq 1
i eq_obj1id neq 1
i eq_obj2id neq 1
b eq_active neq 1
d eq_solref neq 2
d eq_solimp neq 5
d eq_data neq 7
i tendon_adr ntendon 1
i tendon_num ntendon 1
i tendon_matid ntendon 1
i tendon_group ntendon 1
b tendon_limited ntendon 1
d tendon_width ntendon 1
d tendon_solref_lim ntendon 2
d tendon_solimp_lim ntendon 5
d tendon_solref_fri ntendon 2
d tendon_solimp_fri ntendon 5
d tendon_range ntendon 2
d tendon_margin ntendon 1
d tendon_stiffness ntendon 1
d tendon_damping ntendon 1
d tendon_frictionloss ntendon 1
d tendon_lengthspring ntendon 1
d tendon_length0 ntendon 1
d tendon_invweight0 ntendon 1
d tendon_user ntendon nuser_tendon
f tendon_rgba ntendon 4
i wrap_type nwrap 1
i wrap_objid nwrap 1
d wrap_prm nwrap 1
i actuator_trntype nu 1
i actuator_dyntype nu 1
i actuator_gaintype nu 1
i actuator_biastype nu 1
i actuator_trnid nu 2
i actuator_group nu 1
b actuator_ctrllimited nu 1
b actuator_forcelimited nu 1
d actuator_dynprm nu 10
d actuator_gainprm nu 10
d actuator_biasprm nu 10
d actuator_ctrlrange nu 2
d actuator_forcerange nu 2
d actuator_gear nu 6
d actuator_cranklength nu 1
d actuator_acc0 nu 1
d actuator_length0 nu 1
d actuator_lengthrange nu 2
d actuator_user nu nuser_actuator
i sensor_type nsensor 1
i sensor_datatype nsensor 1
i sensor_needstage nsensor 1
i sensor_objtype nsensor 1
i sensor_objid nsensor 1
i sensor_dim nsensor 1
i sensor_adr nsensor 1
d sensor_cutoff nsensor 1
d sensor_noise nsensor 1
d sensor_user nsensor nuser_sensor
i numeric_adr nnumeric 1
i numeric_size nnumeric 1
d numeric_data nnumericdata 1
i text_adr ntext 1
i text_size ntext 1
c text_data ntextdata 1
i tuple_adr ntuple 1
i tuple_size ntuple 1
i tuple_objtype ntupledata 1
i tuple_objid ntupledata 1
d tuple_objprm ntupledata 1
d key_time nkey 1
d key_qpos nkey nq
d key_qvel nkey nv
d key_act nkey na
d key_mpos nkey 3*nmocap
d key_mquat nkey 4*nmocap
i name_bodyadr nbody 1
i name_jntadr njnt 1
i name_geomadr ngeom 1
i name_siteadr nsite 1
i name_camadr ncam 1
i name_lightadr nlight 1
i name_meshadr nmesh 1
i name_skinadr nskin 1
i name_hfieldadr nhfield 1
i name_texadr ntex 1
i name_matadr nmat 1
i name_pairadr npair 1
i name_excludeadr nexclude 1
i name_eqadr neq 1
i name_tendonadr ntendon 1
i name_actuatoradr nu 1
i name_sensoradr nsensor 1
i name_numericadr nnumeric 1
i name_textadr ntext 1
i name_tupleadr ntuple 1
i name_keyadr nkey 1
c names nnames 1
"""

_DTYPES = {"d": np.dtype("<f8"), "i": np.dtype("<i4"), "b": np.dtype("u1"),
           "f": np.dtype("<f4"), "c": np.dtype("S1")}


def _dim(expr: str, sizes: Dict[str, int]) -> int:
    out = 1
    for tok in expr.split("*"):
        out *= int(tok) if tok.isdigit() else sizes[tok]
    return out


class MjbError(ValueError):
    pass


@dataclass
class MjbModel:
    """Decoded model: ``sizes`` / ``opt`` dictionaries and every array by its MuJoCo name."""

    sizes: Dict[str, int]
    opt: Dict[str, object]
    arrays: Dict[str, np.ndarray]
    names: Dict[str, List[str]] = field(default_factory=dict)
    model_name: str = ""
    stat: Dict[str, object] = field(default_factory=dict)

    def __getattr__(self, key):
        d = self.__dict__
        if key in d.get("arrays", {}):
            return d["arrays"][key]
        if key in d.get("sizes", {}):
            return d["sizes"][key]
        raise AttributeError(key)

    def name2id(self, kind: str, name: str) -> int:
        try:
            return self.names[kind].index(name)
        except ValueError as exc:
            raise KeyError(f"no {kind} named {name!r}") from exc


def load_mjb(path: str) -> MjbModel:
    with open(path, "rb") as fh:
        blob = fh.read()
    return parse_mjb(blob)


def parse_mjb(blob: bytes) -> MjbModel:
    if len(blob) < 16 + 4 * N_SIZE_INTS:
        raise MjbError("file too short for an MJB header")
    magic, szf, nint, nptr = struct.unpack_from("<4i", blob, 0)
    if magic != MJB_MAGIC:
        raise MjbError(f"bad magic {magic} (want {MJB_MAGIC})")
    if szf != 8:
        raise MjbError(f"sizeof(mjtNum)={szf}; only double-precision models are supported")
    if nint != N_SIZE_INTS or nptr != N_POINTERS:
        raise MjbError(f"header says {nint} sizes/{nptr} pointers; this reader handles MuJoCo 2.1 "
                       f"({N_SIZE_INTS}/{N_POINTERS})")
    off = 16
    sizes = dict(zip(SIZE_NAMES, struct.unpack_from(f"<{N_SIZE_INTS}i", blob, off)))
    off += 4 * N_SIZE_INTS
    opt: Dict[str, object] = {}
    for name, n in OPT_DOUBLES:
        vals = struct.unpack_from(f"<{n}d", blob, off)
        off += 8 * n
        opt[name] = vals[0] if n == 1 else list(vals)
    for name in OPT_INTS:
        (opt[name],) = struct.unpack_from("<i", blob, off)
        off += 4
    nbuffer = sizes["nbuffer"]
    base = len(blob) - nbuffer
    if base < off:
        raise MjbError("nbuffer larger than the file")
    # mjStatistic {meaninertia, meanmass, meansize, extent, center[3]} closes the fixed-size
    # blocks, immediately before the array buffer (mjVisual sits between mjOption and it).
    if base - off != 608:
        raise MjbError(f"expected 608 bytes of mjVisual+mjStatistic between mjOption and the array buffer "
                       f"(MuJoCo 2.1 layout), found {base - off}: truncated or different version")
    st = struct.unpack_from("<7d", blob, base - 56)
    stat = {"meaninertia": st[0], "meanmass": st[1], "meansize": st[2], "extent": st[3],
            "center": list(st[4:7])}
    arrays: Dict[str, np.ndarray] = {}
    pos = 0
    for line in _SPEC.strip().splitlines():
        t, name, rows, cols = line.split()
        r, c = _dim(rows, sizes), _dim(cols, sizes)
        dt = _DTYPES[t]
        n = r * c
        if n:
            pos = (pos + dt.itemsize - 1) // dt.itemsize * dt.itemsize
        if base + pos + n * dt.itemsize > len(blob):
            raise MjbError(f"array {name} runs past the end of the file")
        a = np.frombuffer(blob, dtype=dt, count=n, offset=base + pos).copy()
        pos += n * dt.itemsize
        arrays[name] = a.reshape(r, c) if c != 1 else a.reshape(r)
    if pos != nbuffer:
        raise MjbError(f"decoded {pos} bytes of arrays but nbuffer={nbuffer}: unsupported layout")
    raw = arrays["names"].tobytes() if sizes["nnames"] else b""

    def _name_at(adr: int) -> str:
        end = raw.find(b"\0", adr)
        return raw[adr:end].decode("utf8", "replace")

    names: Dict[str, List[str]] = {}
    for kind in ("body", "jnt", "geom", "site", "tendon", "actuator", "cam", "light", "mesh",
                 "sensor", "key"):
        adrs = arrays[f"name_{kind}adr"]
        names[kind] = [_name_at(int(a)) for a in adrs]
    # MuJoCo 2.1 stores no separate model-name entry: body 0 ("world") starts at offset 0.
    return MjbModel(sizes=sizes, opt=opt, arrays=arrays, names=names, model_name="", stat=stat)
