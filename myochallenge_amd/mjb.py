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
def _read_layout() -> str:
    """The shared array table (csrc/mjb_layout.inc: also compiled into libmyobatch's C reader)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "mjb_layout.inc")
    with open(path, "r", encoding="utf8") as fh:
        txt = fh.read()
    return txt[txt.index('R"MJBLAYOUT(') + len('R"MJBLAYOUT('):txt.index(')MJBLAYOUT"')]


_SPEC = _read_layout()

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
