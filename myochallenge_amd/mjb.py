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
    neg = [k for k, v in sizes.items() if v < 0]
    if neg:
        raise MjbError(f"negative size(s) in the header: {neg}")
    if len(blob) < off + 8 * sum(n for _, n in OPT_DOUBLES) + 4 * len(OPT_INTS):
        raise MjbError("file too short for mjOption")
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


def dump_mjb(m: MjbModel) -> bytes:
    """Serialises a decoded (or synthetic) model back into the MuJoCo 2.1 ``.mjb`` layout ``parse_mjb`` reads: header, the 57
    sizes, mjOption, 608 bytes of mjVisual + mjStatistic (zeros but for the statistics), then every array of the shared
    layout table in order (arrays the model does not carry are written as zeros of the size the table says).  Used by the
    tests to put models through the C reader (``myo_model_load_mjb``); MuJoCo itself is never involved."""
    sizes = {k: int(m.sizes.get(k, 0)) for k in SIZE_NAMES}
    if "names" not in m.arrays and m.names:               # a synthetic model carries its names as lists: build the name table
        m = MjbModel(sizes=m.sizes, opt=m.opt, arrays=dict(m.arrays), names=m.names, model_name=m.model_name, stat=m.stat)
        table = bytearray()
        for kind, count in (("body", "nbody"), ("jnt", "njnt"), ("geom", "ngeom"), ("site", "nsite"), ("tendon", "ntendon"), ("actuator", "nu")):
            lst = list(m.names.get(kind, [])) + [""] * (sizes[count] - len(m.names.get(kind, [])))
            adr = []
            for nm in lst[:sizes[count]]:
                adr.append(len(table))
                table += nm.encode("utf8") + b"\0"
            m.arrays[f"name_{kind}adr"] = np.array(adr, np.int32)
        m.arrays["names"] = np.frombuffer(bytes(table), dtype="S1")
    rows = []
    pos = 0
    for line in _SPEC.strip().splitlines():
        t, name, r_, c_ = line.split()
        dt = _DTYPES[t]
        a = m.arrays.get(name)
        # sizes that the model does not state follow from the arrays it carries (names, exclude pairs ...)
        if a is not None and not r_.isdigit() and "*" not in r_ and r_ not in m.sizes and r_ != "nbuffer":
            sizes[r_] = max(sizes.get(r_, 0), int(np.asarray(a).shape[0]))
        rows.append((t, name, r_, c_, dt))
    chunks = []
    for t, name, r_, c_, dt in rows:
        n = _dim(r_, sizes) * _dim(c_, sizes)
        a = m.arrays.get(name)
        if a is None:
            raw = np.zeros(n, dt)
        elif t == "c":
            raw = np.frombuffer(np.asarray(a).tobytes(), dtype="S1")
        else:
            raw = np.ascontiguousarray(np.asarray(a)).astype(dt).reshape(-1)
        if raw.size != n:
            raise MjbError(f"array {name} has {raw.size} entries, the sizes say {n}")
        if n:
            pad = -pos % dt.itemsize
            chunks.append(b"\0" * pad)
            pos += pad
        chunks.append(raw.tobytes())
        pos += n * dt.itemsize
    sizes["nbuffer"] = pos
    out = struct.pack("<4i", MJB_MAGIC, 8, N_SIZE_INTS, N_POINTERS) + struct.pack(f"<{N_SIZE_INTS}i", *[sizes[k] for k in SIZE_NAMES])
    o = dict(m.opt)
    defaults = {"apirate": 100.0, "noslip_tolerance": 1e-6, "mpr_tolerance": 1e-6, "wind": [0, 0, 0], "magnetic": [0, -0.5, 0],
                "density": 0.0, "viscosity": 0.0, "o_solref": [0.02, 1.0], "o_solimp": [0.9, 0.95, 0.001, 0.5, 2.0], "o_margin": 0.0,
                "collision": 0, "jacobian": 2, "solver": 2, "noslip_iterations": 0, "mpr_iterations": 50, "enableflags": 0, "disableflags": 0}
    for name, n in OPT_DOUBLES:
        v = o.get(name, defaults.get(name, 0.0))
        out += struct.pack(f"<{n}d", *([v] if n == 1 else list(v)))
    for name in OPT_INTS:
        out += struct.pack("<i", int(o.get(name, defaults.get(name, 0))))
    st = m.stat or {}
    tail = struct.pack("<7d", st.get("meaninertia", 1.0), st.get("meanmass", 1.0), st.get("meansize", 0.1), st.get("extent", 1.0),
                       *st.get("center", [0.0, 0.0, 0.0]))
    out += b"\0" * (608 - len(tail)) + tail
    return out + b"".join(chunks)
