"""Config lowering (reference kwargs -> myo_task_cfg), EnvironmentFactory names, C-ABI surface."""
import ctypes
import json
import os
import re

import pytest

from helpers import make_env

from myochallenge_amd import native
from myochallenge_amd.envs.config import REGISTRATION, make_task_cfg, resolve_kwargs
from myochallenge_amd.model import compile_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_archived_curriculum_config_lowers(golden_dir, models):
    cm = compile_model(models["hand"])
    cfgs = json.load(open(os.path.join(golden_dir, "curriculum_configs.json")))
    assert len(cfgs) == 33
    for path, c in cfgs.items():
        p2 = any(k in c for k in ("task_choice", "obj_size_range", "obj_mass_range", "balls_overlap", "limit_init_angle"))
        name = "CustomMyoBaodingBallsP2" if p2 else "CustomMyoBaodingBallsP1"
        t = make_task_cfg(name, cm, **c)
        assert t.frame_skip == 10 and t.max_episode_steps == 200 and t.n_hand == 23
        w = c["weighted_reward_keys"]
        assert t.weights[0] == w.get("pos_dist_1", 0) and t.weights[5] == w.get("solved", 0)
        if "drop_th" in c:
            assert t.drop_th == c["drop_th"]


def test_registration_defaults(models):
    cm = compile_model(models["hand"])
    p1 = make_task_cfg("CustomMyoBaodingBallsP1", cm)
    assert p1.kind == native.TASK_BAODING_P1 and list(p1.goal_xrange) == [0.025, 0.025] and list(p1.goal_time_period) == [5, 5]
    assert list(p1.weights)[:4] == [5.0, 5.0, 0.0, 0.0] and p1.drop_th == 1.25 and p1.proximity_th == 0.015
    p2 = make_task_cfg("CustomMyoBaodingBallsP2", cm)
    assert p2.kind == native.TASK_BAODING_P2 and p2.task_choice == native.CHOICE_RANDOM
    assert list(p2.goal_time_period) == [4, 6] and list(p2.obj_mass_range) == [0.03, 0.3]
    assert list(p2.obj_friction_change) == [0.2, 0.001, 0.00002] and p2.init_qpos0 == -1.57


def test_bad_arguments_raise_like_the_reference(models):
    cm = compile_model(models["hand"])
    with pytest.raises(ValueError):
        resolve_kwargs("NoSuchEnv")
    with pytest.raises(TypeError):
        make_task_cfg("CustomMyoBaodingBallsP1", cm, not_a_kwarg=1)
    with pytest.raises(ValueError):
        make_task_cfg("CustomMyoBaodingBallsP1", cm, task="sideways")       # baoding.py:296
    with pytest.raises(AssertionError):
        make_task_cfg("CustomMyoBaodingBallsP1", cm, noise_palm=2.0)         # baoding.py:99


def test_environment_factory_names(emu_lib):
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    with pytest.raises(ValueError):
        EnvironmentFactory.create("Nope")
    with pytest.raises(NotImplementedError):
        EnvironmentFactory.create("CustomMyoPenTwirlRandom")          # named by the reference, outside the hot path
    die = make_env("CustomMyoReorientP1", emu_lib, num_envs=2)
    assert die.observation_space.shape == (103,) and die.action_space.shape == (39,)
    die.close()
    env = make_env("CustomMyoBaodingBallsP1", emu_lib, num_envs=2)
    assert env.num_envs == 2 and env.observation_space.shape == (86,) and env.action_space.shape == (39,)
    env.close()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "myobatch.h")).read()
    return sorted(set(re.findall(r"\b(myo_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(native.EXPORTED_SYMBOLS)


@pytest.mark.parametrize("which", ["emu", "hip"])
def test_library_exports_every_declared_symbol(which, emu_lib):
    if which == "emu":
        path = emu_lib.path
    else:
        path = native.LIB_PATH
        if not os.path.exists(path):
            pytest.skip("libmyobatch.so not built in this checkout (run __graft_entry__.build())")
    lib = ctypes.CDLL(path)                   # loading needs libamdhip64 only; no GPU call is made
    # the product exports the whole header; the emulation build (csrc/emu_host.h) the env path, and nothing outside the header
    for sym in (_declared_functions() if which == "hip" else native.ENV_PATH_SYMBOLS):
        assert hasattr(lib, sym), sym
    assert set(native.ENV_PATH_SYMBOLS) <= set(_declared_functions())


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(native.MyoError):
        native.NativeLib(str(tmp_path / "libmyobatch.so"))


def test_task_cfg_struct_matches_header():
    """Field order of the ctypes mirror == field order of myo_task_cfg in the header."""
    src = open(os.path.join(ROOT, "include", "myobatch.h")).read()
    start = src.index("typedef struct myo_task_cfg {") + len("typedef struct myo_task_cfg {")
    body = src[start:src.index("} myo_task_cfg;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        m = re.match(r"(int32_t|double)\s+(.*)", decl, flags=re.S)
        if m:
            names += [re.sub(r"\[.*?\]", "", n).strip() for n in m.group(2).split(",")]
    assert names == [f[0] for f in native.TaskCfg._fields_]


@pytest.mark.parametrize("which", ["emu", "hip"])
def test_c_abi_argument_checks(which, emu_lib):
    """Every entry point validates its arguments before touching a device: NULL handles / pointers and
    non-positive sizes give MYO_E_ARG (-1) with a message, never a crash.  Checked on the PRODUCT library (loading it needs
    libamdhip64 only; an argument error returns before any HIP call) and, for the env path, on the emulation build."""
    import ctypes as C
    if which == "emu":
        lib = emu_lib
    else:
        if not os.path.exists(native.LIB_PATH):
            pytest.skip("libmyobatch.so not built in this checkout (run __graft_entry__.build())")
        lib = native.NativeLib(native.LIB_PATH)
    L = lib.L
    null = None
    f = C.c_float
    checks = [
        (L.myo_model_from_blob, (null, 0, C.byref(C.c_void_p()))),
        (L.myo_batch_create, (null, null, 4, 0, 0, 1, C.byref(C.c_void_p()))),
        (L.myo_batch_reset, (null, null, null, null)),
        (L.myo_batch_step, (null,) * 10),
        (L.myo_batch_step_inner, (null,) * 6),
        (L.myo_batch_physics_step, (null, null, 1, null)),
        (L.myo_batch_get_state, (null,) * 6),
        (L.myo_batch_set_task, (null,) * 5),
        (L.myo_batch_set_object_group, (null, 0, 1)),
        (L.myo_batch_bind_constants, (null, null)),
        (L.myo_batch_tune_wrap_order, (null, null)),
    ]
    if which == "hip":
        checks += [
            (L.myo_ppo_loss_grad, (null,) * 8 + (16, 39, f(0.2), f(0.5)) + (null,) * 6 + (0, f(0.0)) + (null,) * 4),
            (L.myo_ppo_gather, (null,) * 6 + (16, 86, 39, null, 2) + (null,) * 7),
            (L.myo_bias_relu_bf16, (null, null, 2, 16, 256, null)),
            (L.myo_relu_bwd_colsum_bf16, (null, null, 64, 256, null, null)),
            (L.myo_splitk_reduce, (null, 1, null, 2, 32, 256, null)),
            (L.myo_splitk_reduce2, (null, 1, null, 2, 32, 256, null, 0, null, 2, 64, 256, null)),
            (L.myo_adam_clip_step, (null,) * 4 + (10,) + (f(0.1),) * 6 + (null,) * 4),
            (L.myo_gae, (null,) * 5 + (4, 4, f(0.99), f(0.95), null, null, null)),
            (L.myo_rollout_policy_input, (null, 4, 86, null, null, 2, null, null)),
            (L.myo_rollout_sample, (null, null, null, 4, 39, 1, null, null, null, null, null, null, 0, null)),
            (L.myo_vecnorm_step, (null,) * 5 + (4, 86) + (null,) * 5 + (0.99, 1e-8, 10.0, 10.0, 1, 1, 1) + (null,) * 9),
            (L.myo_rollout_advance, (null, 4, null, null)),
            (L.myo_lstm_cell_fwd, (null,) * 4 + (8, 4, 16, 0) + (null,) * 6),
            (L.myo_lstm_cell_bwd, (null,) * 7 + (8, 4, 16, 0) + (null,) * 3),
            (L.myo_lstm_step_fwd, (null, 0, 0) + (null,) * 4 + (2, 16, 32, null, 0) + (null,) * 7),
            (L.myo_lstm_step_bwd, (null, 0) + (null,) * 7 + (2, 16, 32) + (null,) * 3),
            (L.myo_lstm_seq_fwd, (null, 0, 0, 0) + (null,) * 4 + (2, 16, 128, 4, 1, null, 0, 0) + (null,) * 4),
            (L.myo_lstm_seq_bwd, (null, 0, 0) + (null,) * 5 + (2, 16, 128, 4, 1) + (null,) * 2),
        ]
    for fn, args in checks:
        rc = fn(*args)
        assert rc == -1, (fn.__name__, rc)
        assert len(L.myo_last_error()) > 0
    assert L.myo_batch_num_envs(null) <= 0 and L.myo_batch_obs_dim(null) <= 0
    if which == "hip":
        buf = (C.c_float * 64)()
        assert L.myo_splitk_reduce(buf, 0, buf, 1, 2, 15, None) == -1          # odd n


def test_build_identity_covers_every_included_file():
    """source_id() / _stale() must see every file the library is compiled from."""
    from myochallenge_amd import build
    listed = {os.path.normpath(os.path.join(build.CSRC, s)) for s in build.SOURCES} | {os.path.normpath(h) for h in build.HEADERS}
    missing = build.reachable_includes() - listed
    assert not missing, f"not in build.SOURCES/HEADERS: {sorted(missing)}"


def test_bench_gpus_n_spawns_a_launcher_before_torch(monkeypatch):
    """`python bench.py --gpus 4` with no WORLD_SIZE must start 4 ranks (torch.distributed.run on 127.0.0.1) as a child and
    return its exit code — without importing torch in the parent."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
