"""An env step run in parts (k_step's launch plan: substeps 4 + 3 + 2 + 1 handed from workgroup to workgroup through the env
record) must give the BITS of the whole step: same observations, rewards, dones, terminal observations and states, for both
steppers and both integrators, through resets and P2's per-episode ball draws.  The emulation build runs the parts one after
the other through the record (same load_env / store_env code); the GPU test adds the claim / publish protocol of the real
launch (tests of the physics itself: test_emu_parity.py, test_gpu_parity.py)."""
import os

import numpy as np
import pytest

from helpers import Mem
from myochallenge_amd import native
from myochallenge_amd.envs.config import make_task_cfg
from myochallenge_amd.model import compile_model


def _rollout(lib, cm, env_name, dtype, n, nsteps, split, order=None, seed=7, horizon=4, generation=None, publish=None, overflow_ok=False, wrap_order=True, tune_every=0):
    """Observations, rewards, dones, terminal observations of `nsteps` steps + the final state, with MYO_STEP_SPLIT = split."""
    old = {k: os.environ.get(k) for k in ("MYO_STEP_SPLIT", "MYO_STEP_ORDER", "MYO_PUBLISH", "MYO_NO_WRAP_ORDER")}
    try:
        if wrap_order:
            os.environ.pop("MYO_NO_WRAP_ORDER", None)
        else:
            os.environ["MYO_NO_WRAP_ORDER"] = "1"       # the wraps stay in the model's order (A/B switch, include/myobatch.h)
        if split is None:
            os.environ.pop("MYO_STEP_SPLIT", None)
        else:
            os.environ["MYO_STEP_SPLIT"] = split
        if order is None:
            os.environ.pop("MYO_STEP_ORDER", None)
        else:
            os.environ["MYO_STEP_ORDER"] = order
        if publish is None:
            os.environ.pop("MYO_PUBLISH", None)
        else:
            os.environ["MYO_PUBLISH"] = publish     # "fence": round 4's agent release fence instead of write-through records
        mem = Mem(lib)
        if env_name.startswith("CustomMyoReorient"):
            from myochallenge_amd.envs.reorient import make_reorient_cfg
            tc = make_reorient_cfg(env_name, cm, max_episode_steps=horizon)
        else:
            tc = make_task_cfg(env_name, cm, drop_th=1.3, max_episode_steps=horizon)
        b = native.Batch(native.Model(cm, lib), tc, n, 0, seed, dtype)       # the plan is read when the batch is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if generation is not None:
        lib.check(lib.L.myo_batch_set_step_generation(b.h, generation))
    nobs = b.obs_dim
    obs = mem.zeros((n, nobs), np.float32)
    rew, done, trunc = mem.zeros(n, np.float32), mem.zeros(n, np.uint8), mem.zeros(n, np.uint8)
    term, comps, ep = mem.zeros((n, nobs), np.float32), mem.zeros((n, 8), np.float32), mem.zeros((n, 2), np.float32)
    b.reset(None, obs)
    rng = np.random.RandomState(3)
    out = [mem.host(obs).copy()]
    for i in range(nsteps):
        a = np.clip(rng.normal(0, 0.5, (n, 39)), -1.2, 1.2).astype(np.float32)
        if tune_every and i % tune_every == 0:
            b.tune_wrap_order()
        b.step(mem.arr(a, np.float32), obs, rew, done, trunc, term, comps, ep)
        out += [mem.host(x).copy() for x in (obs, rew, done, trunc, term, comps, ep)]
    qpos, qvel, act, time = mem.zeros((n, cm.size("nq"))), mem.zeros((n, cm.size("nv"))), mem.zeros((n, cm.size("na"))), mem.zeros(n)
    b.get_state(qpos, qvel, act, time)
    out += [mem.host(x).copy() for x in (qpos, qvel, act, time)]
    h = b.health()
    assert h["protocol_errors"] == 0 and h["limit_row_overflows"] == 0 and (overflow_ok or h["contact_overflows"] == 0), h     # no hand-off state of another generation, no dropped contact
    b.close()
    return out


def _same_bits(a, b):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.dtype == y.dtype and x.tobytes() == y.tobytes(), f"array {i} differs (max abs diff {np.abs(x.astype(float) - y.astype(float)).max()})"


@pytest.mark.parametrize("dtype", [native.MYO_F64, native.MYO_MIXED], ids=["f64", "mixed"])
@pytest.mark.parametrize("integ", [0, 1], ids=["euler", "rk4"])
def test_step_parts_give_the_whole_steps_bits_on_emulation(emu_lib, models, dtype, integ):
    cm = compile_model(models["hand"], integrator=integ)
    n, steps = 3, (14 if integ == 0 else 5)
    whole = _rollout(emu_lib, cm, "CustomMyoBaodingBallsP2", dtype, n, steps, "0")
    _same_bits(whole, _rollout(emu_lib, cm, "CustomMyoBaodingBallsP2", dtype, n, steps, None))        # the default plan
    if integ == 0:
        _same_bits(whole, _rollout(emu_lib, cm, "CustomMyoBaodingBallsP2", dtype, n, steps, "7,3"))
        _same_bits(whole, _rollout(emu_lib, cm, "CustomMyoBaodingBallsP2", dtype, n, steps, "1,1,1,1,6"))
    assert any(w.dtype == np.uint8 and w.any() for w in whole), "the rollout should cross an episode end (TimeLimit 4)"


def test_step_parts_of_the_die_on_emulation(emu_lib):
    """The die on the fp64 stepper (48-slot scratch: contact records and wrap results in the wave slot's global workspace, activations and object friction read and written IN the record, tendon lengths /
    activation rates / reward terms in the env workspace — Scratch::SPILL) in parts = whole steps, through a reset."""
    from myochallenge_amd.synth_hand import synthetic_hand_die
    die = compile_model(synthetic_hand_die(), integrator=0, unsupported_contacts="drop")
    w = _rollout(emu_lib, die, "CustomMyoReorientP2", native.MYO_F64, 2, 7, "0", horizon=3, overflow_ok=True)
    _same_bits(w, _rollout(emu_lib, die, "CustomMyoReorientP2", native.MYO_F64, 2, 7, None, horizon=3, overflow_ok=True))
    _same_bits(w, _rollout(emu_lib, die, "CustomMyoReorientP2", native.MYO_F64, 2, 7, "1,1,3", horizon=3, overflow_ok=True))
    assert any(x.dtype == np.uint8 and x.any() for x in w)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [native.MYO_F64, native.MYO_MIXED], ids=["f64", "mixed"])
def test_step_parts_give_the_whole_steps_bits_on_gpu(hip_lib, models, dtype):
    """512 envs (more workgroups than one part's share of the slots is not needed: the protocol is per env), 30 steps; the
    opt-in launch order on top."""
    cm = compile_model(models["hand"], integrator=0)
    whole = _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", dtype, 512, 30, "0")
    _same_bits(whole, _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", dtype, 512, 30, None))
    _same_bits(whole, _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", dtype, 512, 30, "5,3,1,1", order="1"))
    assert any(w.dtype == np.uint8 and w.any() for w in whole)


@pytest.mark.gpu
def test_step_parts_rk4_and_full_batch_on_gpu(hip_lib, models):
    """RK4 (its stage storage is per workgroup) and a batch that fills the chip twice: 4096 envs, every slot busy, parts of
    different envs interleaved on every CU."""
    cm = compile_model(models["hand"], integrator=1)
    _same_bits(_rollout(hip_lib, cm, "CustomMyoBaodingBallsP1", native.MYO_MIXED, 256, 6, "0"),
               _rollout(hip_lib, cm, "CustomMyoBaodingBallsP1", native.MYO_MIXED, 256, 6, None))
    cm = compile_model(models["hand"], integrator=0)
    _same_bits(_rollout(hip_lib, cm, "CustomMyoBaodingBallsP1", native.MYO_MIXED, 4096, 12, "0"),
               _rollout(hip_lib, cm, "CustomMyoBaodingBallsP1", native.MYO_MIXED, 4096, 12, None))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [native.MYO_F64, native.MYO_MIXED], ids=["f64", "mixed"])
def test_wrap_order_changes_no_bit_on_gpu(hip_lib, models, dtype):
    """The tendon stage's geom wraps are re-sorted by how often they engage (a census of the envs' states after a reset of all envs,
    16 steps later, then every 256 steps, or when the caller asks: myo_batch_tune_wrap_order), so that the wraps that rarely engage
    share the solver's last pass and that pass skips its tangent solve.  Each wrap's arithmetic is its own: a rollout with the
    model's order (MYO_NO_WRAP_ORDER), one with the automatic census and one re-sorted every third step give the same bits — the
    hand has 71 geom wraps (two passes), 24 steps cross the 16-step census and episode ends."""
    cm = compile_model(models["hand"], integrator=0)
    fixed = _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", dtype, 256, 24, None, wrap_order=False)
    _same_bits(fixed, _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", dtype, 256, 24, None))
    _same_bits(fixed, _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", dtype, 256, 24, None, tune_every=3))
    assert any(w.dtype == np.uint8 and w.any() for w in fixed)


@pytest.mark.gpu
def test_workspace_blocks_are_returned_on_gpu(hip_lib, models):
    """The fp64 stepper's workspace blocks are OWNED by a workgroup while it holds an env (csrc/wave.h: myo_ws_acquire / myo_ws_release)
    — not implied by the hardware wave slot, which a wavefront does not keep for life.  Every entry point that loads an env must give
    its block back: 4096 workgroups a launch against 2048 blocks per XCD would exhaust the pool within two launches of a leaking path,
    and an exhausted pool is counted in health[0]."""
    import torch
    cm = compile_model(models["hand"], integrator=0)
    tc = make_task_cfg("CustomMyoBaodingBallsP2", cm, drop_th=1.3, max_episode_steps=6)
    n = 4096
    b = native.Batch(native.Model(cm, hip_lib), tc, n, 0, 3, native.MYO_F64)
    dev = torch.device("cuda:0")
    obs = torch.zeros((n, b.obs_dim), dtype=torch.float32, device=dev)
    rew, done = torch.zeros(n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
    act = torch.zeros((n, 39), dtype=torch.float32, device=dev)
    ctrl = torch.full((n, 39), 0.2, dtype=torch.float64, device=dev)
    dump = torch.zeros((n, b.dump_size), dtype=torch.float64, device=dev)
    mask = torch.ones(n, dtype=torch.uint8, device=dev)
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    b.reset(None, obs)
    for _ in range(6):
        b.step(act, obs, rew, done)                       # (the step plan: four workgroups an env)
        b.step_inner(mask, act, obs, done)
        b.step_inner_idx(idx, act, obs, done)
        b.physics_step(ctrl, 2)
        b.forward_dump(ctrl, dump)
        b.tune_wrap_order()
        b.reset(mask, obs)
    torch.cuda.synchronize()
    h = b.health()
    assert h["protocol_errors"] == 0, h
    b.close()


@pytest.mark.gpu
def test_step_plan_generation_counter_wraps(hip_lib, models):
    """The protocol's state is 16 x generation + part: the counter wraps after 2^28 steps (days of stepping); the arithmetic is
    modulo 2^32 and a rollout that crosses the wrap gives the same bits."""
    cm = compile_model(models["hand"], integrator=0)
    whole = _rollout(hip_lib, cm, "CustomMyoBaodingBallsP1", native.MYO_MIXED, 256, 8, "0")
    _same_bits(whole, _rollout(hip_lib, cm, "CustomMyoBaodingBallsP1", native.MYO_MIXED, 256, 8, None, generation=2 ** 28 - 3))
    _same_bits(whole, _rollout(hip_lib, cm, "CustomMyoBaodingBallsP1", native.MYO_MIXED, 256, 8, None, generation=2 ** 32 - 3))


@pytest.mark.gpu
def test_step_parts_publish_forms_and_the_die_on_gpu(hip_lib, models):
    """The two ways a part hands its record on — write-through stores (the default) and round 4's agent release fence
    (MYO_PUBLISH=fence) — give the bits of whole steps; and so does the die on the fp64 stepper, whose 48-slot scratch reads and
    writes the activations and the object friction IN the record and keeps its tendon lengths / activation rates in the env
    workspace (Scratch::SPILL): frame_skip 5 = parts of 2 + 1 + 1 + 1 substeps."""
    from myochallenge_amd.synth_hand import synthetic_hand_die
    cm = compile_model(models["hand"], integrator=0)
    whole = _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", native.MYO_F64, 512, 20, "0")
    _same_bits(whole, _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", native.MYO_F64, 512, 20, None, publish="fence"))
    _same_bits(whole, _rollout(hip_lib, cm, "CustomMyoBaodingBallsP2", native.MYO_F64, 512, 20, None, publish="wt"))
    die = compile_model(synthetic_hand_die(), integrator=0, unsupported_contacts="drop")
    for dtype in (native.MYO_F64, native.MYO_MIXED):
        w = _rollout(hip_lib, die, "CustomMyoReorientP2", dtype, 512, 16, "0", horizon=6, overflow_ok=True)
        _same_bits(w, _rollout(hip_lib, die, "CustomMyoReorientP2", dtype, 512, 16, None, horizon=6, overflow_ok=True))
        _same_bits(w, _rollout(hip_lib, die, "CustomMyoReorientP2", dtype, 512, 16, None, horizon=6, publish="fence", overflow_ok=True))
        assert any(x.dtype == np.uint8 and x.any() for x in w)


@pytest.mark.gpu
def test_wave_slots_are_exclusive_on_gpu(hip_lib):
    """What k_step's per-wave-slot workspace rests on (csrc/wave.h: myo_wave_slot = XCC | SE | SH | CU | SIMD | wave buffer from
    HW_REG_XCC_ID / HW_REG_HW_ID): no two workgroups that are resident at the same time decode the same slot.  16,384 one-wave
    workgroups with k_step's LDS footprint (eight per CU: the chip is full for eight rounds) each hold their slot's counter for
    ~20 us; none may find it taken, the slots seen are as many as the chip holds at that residency, and all eight XCDs appear."""
    import ctypes as C
    out = (C.c_int32 * 4)()
    hip_lib.check(hip_lib.L.myo_debug_wave_slots(0, 16384, 20320, out))
    clashes, distinct, top, xcc_mask = (int(v) for v in out)
    assert clashes == 0, (clashes, distinct)
    assert 256 <= distinct <= 8 * 40 * 4 * 10 and top < 8 * 16384, (distinct, top)
    assert xcc_mask == 0xff, hex(xcc_mask)
    # one wave per SIMD (a 64 KB footprint): still exclusive
    hip_lib.check(hip_lib.L.myo_debug_wave_slots(0, 4096, 65536, out))
    assert int(out[0]) == 0, tuple(out)
