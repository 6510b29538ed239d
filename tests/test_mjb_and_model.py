"""MJB reader, model compilation, blob format, host-side constants (CPU only)."""
import json
import os
import struct

import numpy as np
import pytest

from myochallenge_amd.mjb import MjbError, load_mjb, parse_mjb
from myochallenge_amd.model import ModelError, compile_model
from myochallenge_amd.setconst import set_const


@pytest.mark.parametrize("name,short", [("myo_finger_v0", "finger"), ("motor_finger_v0", "motor_finger"), ("myo_load", "load")])
def test_mjb_decodes_to_committed_json(golden_dir, name, short):
    m = load_mjb(os.path.join(golden_dir, name + ".mjb"))
    js = json.load(open(os.path.join(golden_dir, f"mjb_{short}.json")))
    assert m.sizes == js["sizes"]
    assert m.opt["timestep"] == 0.002 and m.opt["integrator"] == 0 and m.opt["cone"] == 0
    for k, v in js["arrays"].items():
        np.testing.assert_array_equal(np.asarray(m.arrays[k]).reshape(-1), np.asarray(v).reshape(-1))


def test_mjb_known_facts(models):
    f = models["finger"]
    assert (f.nq, f.nv, f.nu, f.ntendon, f.nwrap) == (4, 4, 5, 5, 35)
    assert f.names["tendon"] == ["extn", "mflx", "dflx", "adabR", "adabL"]
    assert abs(f.stat["meaninertia"] - np.mean(f.dof_M0)) < 1e-15
    l = models["load"]
    assert (l.nq, l.nu, l.ntendon) == (1, 1, 1) and l.tendon_length0[0] == 0.1


def test_mjb_rejects_garbage(golden_dir):
    blob = open(os.path.join(golden_dir, "myo_load.mjb"), "rb").read()
    with pytest.raises(MjbError):
        parse_mjb(b"\0" * 10)
    with pytest.raises(MjbError):
        parse_mjb(struct.pack("<4i", 1, 8, 57, 266) + blob[16:])
    with pytest.raises(MjbError):
        parse_mjb(blob[:-7])       # truncated buffer


def test_blob_roundtrip(models):
    cm = compile_model(models["finger"], unsupported_contacts="drop")
    blob = cm.to_blob()
    magic, ver, nf, tot = struct.unpack_from("<IIII", blob, 0)
    assert magic == 0x4D4F594D and ver == 1 and tot == len(blob) and nf == len(cm.fields)
    # every field is 8-byte aligned and inside the blob
    for i in range(nf):
        name, dt, cnt, off = struct.unpack_from("<40sIIQ", blob, 16 + 56 * i)
        assert off % 8 == 0 and off + cnt * (4 if dt == 0 else 8) <= len(blob)


def test_collision_pair_filter(models):
    cm = compile_model(models["hand"])
    g1, g2 = cm.x_pair_geom1, cm.x_pair_geom2
    names = cm.names["geom"]
    # only ball-vs-hand and ball-vs-ball pairs (hand geoms have conaffinity 0)
    assert all("ball" in names[a] or "ball" in names[b] for a, b in zip(g1, g2))
    assert sum(1 for a, b in zip(g1, g2) if "ball" in names[a] and "ball" in names[b]) == 1
    assert cm.dropped_pairs == []
    # finger model: cylinder / ellipsoid pairs make the model an error unless the caller opts in, and are then listed
    import pytest
    from myochallenge_amd.model import UnsupportedContactsError
    with pytest.raises(UnsupportedContactsError):
        compile_model(models["finger"])
    assert len(compile_model(models["finger"], unsupported_contacts="drop").dropped_pairs) > 0


def test_feature_gates(models):
    import copy
    m = copy.deepcopy(models["finger"])
    m.arrays["dof_frictionloss"][0] = 0.1
    with pytest.raises(ModelError):
        compile_model(m)


def test_setconst_reproduces_mujoco_constants(models):
    import copy
    m = copy.deepcopy(models["finger"])
    ref = {k: m.arrays[k].copy() for k in ("dof_M0", "dof_invweight0", "body_invweight0", "tendon_length0")}
    refJ = {k: m.arrays[k].copy() for k in ("tendon_invweight0", "actuator_acc0")}
    set_const(m)
    for k, v in ref.items():
        np.testing.assert_allclose(m.arrays[k], v, rtol=1e-12, atol=1e-15)
    for k, v in refJ.items():   # finite-difference moment arms vs MuJoCo's analytic ones
        np.testing.assert_allclose(m.arrays[k], v, rtol=1e-4)


def test_synthetic_hand_shape(models):
    h = models["hand"]
    assert (h.nq, h.nv, h.nu, h.na) == (37, 35, 39, 39)          # baoding.py:183,187-194,282
    assert h.names["site"][h.name2id("site", "target1_site")] == "target1_site"
    jn = h.names["jnt"]
    assert jn[0] == "pro_sup" and jn[3].startswith("cmc") and jn[8].endswith("abduction") and jn[12].endswith("abduction")
    assert h.opt["timestep"] == 0.002 and h.opt["integrator"] == 0
