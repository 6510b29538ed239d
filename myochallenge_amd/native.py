"""ctypes binding of libmyobatch (C ABI: include/myobatch.h).

The product library is ``myochallenge_amd/libmyobatch.so`` built by ``__graft_entry__.build()``
with hipcc for gfx950.  There is no CPU fallback: if the library is missing, ``load()`` raises.
(``tests/emu`` builds a lane-serial emulation of the same kernel source for debugging; tests
load it by passing its explicit path — nothing in the package ever does.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmyobatch.so")

MYO_F64, MYO_MIXED = 0, 1     # stepper arithmetic: all fp64 | mixed (fp64 state, kinematic chain, contact distances; fp32 dynamics)
MYO_F32 = MYO_MIXED             # the name of round 1, when dtype 1 was a pure fp32 stepper
TASK_NONE, TASK_BAODING_P1, TASK_BAODING_P2, TASK_REORIENT = 0, 1, 2, 3
CHOICE_FIXED, CHOICE_CW, CHOICE_CCW, CHOICE_RANDOM = 0, 1, 2, 3
N_RWD = 8
RWD_KEYS = ("pos_dist_1", "pos_dist_2", "act_reg", "alive", "sparse", "solved", "done", "dense")


ROT_CHOICE_MAX, OBJG_MAX = 4, 20      # include/myobatch.h


class TaskCfg(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("frame_skip", C.c_int32), ("max_episode_steps", C.c_int32),
        ("n_hand", C.c_int32),
        ("obj1_sid", C.c_int32), ("obj2_sid", C.c_int32), ("target1_sid", C.c_int32),
        ("target2_sid", C.c_int32), ("obj1_bid", C.c_int32), ("obj2_bid", C.c_int32),
        ("obj1_gid", C.c_int32), ("obj2_gid", C.c_int32),
        ("task_choice", C.c_int32), ("enable_rsi", C.c_int32), ("balls_overlap", C.c_int32),
        ("limit_init_angle_on", C.c_int32), ("beta_init_angle_on", C.c_int32),
        ("beta_ball_size_on", C.c_int32), ("beta_ball_mass_on", C.c_int32),
        ("drop_th", C.c_double), ("proximity_th", C.c_double), ("center_pos", C.c_double * 2),
        ("weights", C.c_double * 7),
        ("goal_time_period", C.c_double * 2), ("goal_xrange", C.c_double * 2),
        ("goal_yrange", C.c_double * 2),
        ("rsi_probability", C.c_double), ("overlap_probability", C.c_double),
        ("noise_palm", C.c_double), ("noise_fingers", C.c_double), ("noise_balls", C.c_double),
        ("limit_init_angle", C.c_double), ("beta_init_angle", C.c_double * 2),
        ("beta_ball_size", C.c_double * 2), ("beta_ball_mass", C.c_double * 2),
        ("obj_size_range", C.c_double * 2), ("obj_mass_range", C.c_double * 2),
        ("obj_friction_change", C.c_double * 3), ("init_qpos0", C.c_double),
        # die reorient (kind TASK_REORIENT)
        ("ro_weights", C.c_double * 9), ("ro_goal_pos", C.c_double * 2), ("ro_goal_rot", C.c_double * 2),
        ("ro_n_rot_choice", C.c_int32 * 3), ("ro_pad_", C.c_int32),
        ("ro_rot_choice", C.c_double * 2 * ROT_CHOICE_MAX * 3),
        ("ro_obj_size_change", C.c_double), ("ro_pos_th", C.c_double), ("ro_rot_th", C.c_double),
        ("ro_goal_init_pos", C.c_double * 3), ("ro_goal_obj_offset", C.c_double * 3),
        ("ro_rsi_distance_pos", C.c_double), ("ro_rsi_distance_rot", C.c_double),
    ]


class PpoMlpDesc(C.Structure):
    """myo_ppo_mlp_desc of include/myobatch.h (one fused PPO minibatch step of the MLP actor-critic)."""
    _fields_ = [
        ("obs", C.c_void_p), ("act", C.c_void_p), ("oldlp", C.c_void_p), ("adv", C.c_void_p), ("ret", C.c_void_p), ("idx", C.c_void_p),
        ("B", C.c_int32), ("O", C.c_int32), ("A", C.c_int32), ("hidden", C.c_int32),
        ("params", C.c_void_p), ("grads", C.c_void_p), ("G", C.c_int64),
        ("off_W1", C.c_int64 * 2), ("off_b1", C.c_int64 * 2), ("off_W2", C.c_int64 * 2), ("off_b2", C.c_int64 * 2),
        ("off_Wh", C.c_int64 * 2), ("off_bh", C.c_int64 * 2), ("off_log_std", C.c_int64),
        ("clip", C.c_float), ("vf_coef", C.c_float), ("ent_coef", C.c_float), ("compute_adv_stats", C.c_int32),
        ("adv_stats", C.c_void_p), ("acc", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("sqnorm_part", C.c_void_p), ("adam_step", C.c_void_p),
    ]


class PpoMlpRolloutDesc(C.Structure):
    """myo_ppo_mlp_rollout_desc of include/myobatch.h (one env step's policy call of the rollout, MLP actor-critic)."""
    _fields_ = [
        ("obs", C.c_void_p), ("N", C.c_int32), ("O", C.c_int32), ("A", C.c_int32), ("hidden", C.c_int32), ("params", C.c_void_p),
        ("off_W1", C.c_int64 * 2), ("off_b1", C.c_int64 * 2), ("off_W2", C.c_int64 * 2), ("off_b2", C.c_int64 * 2),
        ("off_Wh", C.c_int64 * 2), ("off_bh", C.c_int64 * 2), ("off_log_std", C.c_int64),
        ("seed", C.c_uint64), ("draw_counter", C.c_void_p), ("t_idx", C.c_void_p),
        ("obs_buf", C.c_void_p), ("act_buf", C.c_void_p), ("val_buf", C.c_void_p), ("logp_buf", C.c_void_p), ("clipped", C.c_void_p),
        ("deterministic", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
    ]


class MyoError(RuntimeError):
    pass


class NativeLib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise MyoError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()').  libmyobatch has no CPU fallback.")
        L = C.CDLL(path)
        self.path = path
        self.L = L
        vp, i32, u64, dbl = C.c_void_p, C.c_int, C.c_uint64, C.c_double
        L.myo_last_error.restype = C.c_char_p
        L.myo_version.restype = C.c_char_p
        L.myo_model_from_blob.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.myo_model_load_mjb.argtypes = [C.c_char_p, i32, i32, C.POINTER(vp)]
        L.myo_model_destroy.argtypes = [vp]
        L.myo_model_size.argtypes = [vp, C.c_char_p]
        L.myo_batch_create.argtypes = [vp, C.POINTER(TaskCfg), i32, i32, u64, i32, C.POINTER(vp)]
        L.myo_batch_destroy.argtypes = [vp]
        L.myo_batch_num_envs.argtypes = [vp]
        L.myo_batch_set_step_generation.argtypes = [vp, C.c_uint32]
        L.myo_batch_health.argtypes = [vp, C.POINTER(i32)]
        L.myo_debug_wave_slots.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(i32)]
        L.myo_batch_obs_dim.argtypes = [vp]
        L.myo_batch_lds_bytes.argtypes = [vp]
        L.myo_batch_reset.argtypes = [vp, vp, vp, vp]
        L.myo_batch_step.argtypes = [vp] * 10
        L.myo_batch_step_inner.argtypes = [vp] * 6
        L.myo_batch_step_inner_idx.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        L.myo_batch_copy_envs.argtypes = [vp, vp, vp, vp, i32, vp]
        L.myo_batch_physics_step.argtypes = [vp, vp, i32, vp]
        L.myo_batch_get_state.argtypes = [vp] * 6
        L.myo_batch_set_state.argtypes = [vp] * 6
        L.myo_batch_set_task.argtypes = [vp] * 5
        L.myo_batch_warmstart.argtypes = [vp] * 4
        L.myo_batch_set_bad_state_buffer.argtypes = [vp, vp]
        L.myo_batch_get_task.argtypes = [vp] * 5
        L.myo_batch_bind_constants.argtypes = [vp, vp]
        L.myo_batch_tune_wrap_order.argtypes = [vp, vp]
        L.myo_batch_set_object_group.argtypes = [vp, i32, i32]
        L.myo_batch_object_friction.argtypes = [vp, vp, vp, vp]
        L.myo_batch_forward_dump.argtypes = [vp, vp, vp, vp]
        L.myo_batch_dump_size.argtypes = [vp]
        L.myo_batch_dump_offset.argtypes = [vp, C.c_char_p]
        L.myo_batch_kernel_ms.argtypes = [vp]
        L.myo_batch_kernel_ms.restype = dbl
        L.myo_batch_enable_timing.argtypes = [vp, i32]
        if b"MYO_EMU" in L.myo_version():
            return      # the emulation build (test tooling, csrc/emu_host.h) implements the env path only: ENV_PATH_SYMBOLS
        L.myo_ppo_loss_grad.argtypes = [vp] * 8 + [i32, i32, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, i32, C.c_float, vp, vp, vp, vp]
        L.myo_ppo_gather.argtypes = [vp] * 6 + [i32, i32, i32, vp, i32] + [vp] * 7
        L.myo_bias_relu_bf16.argtypes = [vp, vp, i32, i32, i32, vp]
        L.myo_splitk_reduce.argtypes = [vp, i32, vp, i32, i32, i32, vp]
        L.myo_splitk_reduce2.argtypes = [vp, i32, vp, i32, i32, i32, vp, i32, vp, i32, i32, i32, vp]
        L.myo_relu_bwd_colsum_bf16.argtypes = [vp, vp, i32, i32, vp, vp]
        L.myo_rollout_policy_input.argtypes = [vp, i32, i32, vp, vp, i32, vp, vp]
        L.myo_rollout_sample.argtypes = [vp, vp, vp, i32, i32, u64, vp, vp, vp, vp, vp, vp, i32, vp]
        L.myo_vecnorm_step.argtypes = [vp] * 5 + [i32, i32] + [vp] * 5 + [dbl] * 4 + [i32] * 3 + [vp] * 9
        L.myo_rollout_advance.argtypes = [vp, i32, vp, vp]
        L.myo_rollout_sample_sde.argtypes = [vp] * 4 + [i32] * 3 + [vp] * 3 + [i32, vp]
        L.myo_vecnorm_batch_moments.argtypes = [vp, vp, i32, i32, vp, dbl, i32, vp, vp, vp]
        L.myo_vecnorm_finish.argtypes = [vp] * 5 + [i32, i32] + [vp] * 5 + [dbl] * 3 + [i32] * 3 + [vp] * 9
        L.myo_gae.argtypes = [vp] * 5 + [i32, i32, C.c_float, C.c_float, vp, vp, vp]
        L.myo_lstm_cell_fwd.argtypes = [vp] * 4 + [i32] * 4 + [vp] * 6
        L.myo_lstm_cell_bwd.argtypes = [vp] * 7 + [i32] * 4 + [vp] * 3
        L.myo_lstm_step_supported.argtypes = [i32]
        L.myo_lstm_step_fwd.argtypes = [vp, C.c_longlong, C.c_longlong] + [vp] * 4 + [i32] * 3 + [vp, C.c_longlong] + [vp] * 7
        L.myo_lstm_step_bwd.argtypes = [vp, C.c_longlong] + [vp] * 7 + [i32] * 3 + [vp] * 3
        L.myo_lstm_seq_supported.argtypes = [i32]
        L.myo_lstm_seq_fwd.argtypes = [vp] + [C.c_longlong] * 3 + [vp] * 4 + [i32] * 5 + [vp, C.c_longlong, C.c_longlong] + [vp] * 4
        L.myo_lstm_seq_bwd.argtypes = [vp, C.c_longlong, C.c_longlong] + [vp] * 5 + [i32] * 5 + [vp] * 2
        L.myo_adam_clip_step.argtypes = [vp] * 4 + [i32] + [C.c_float] * 6 + [vp, vp, vp, vp]
        L.myo_adam_apply.argtypes = [vp] * 4 + [i32] + [C.c_float] * 6 + [vp, vp, i32, vp, vp]
        L.myo_ppo_mlp_workspace_bytes.restype = C.c_longlong
        L.myo_ppo_mlp_workspace_bytes.argtypes = [i32, i32, i32, i32, C.c_longlong]
        L.myo_ppo_mlp_step.argtypes = [C.POINTER(PpoMlpDesc), vp]
        L.myo_ppo_mlp_sqnorm_parts.argtypes = [i32]
        L.myo_ppo_mlp_rollout_workspace_bytes.restype = C.c_longlong
        L.myo_ppo_mlp_rollout_workspace_bytes.argtypes = [i32, i32, i32]
        L.myo_ppo_mlp_rollout_refresh.argtypes = [C.POINTER(PpoMlpRolloutDesc), vp]
        L.myo_ppo_mlp_rollout.argtypes = [C.POINTER(PpoMlpRolloutDesc), vp]

    def check(self, rc: int):
        if rc != 0:
            raise MyoError(f"libmyobatch error {rc}: {self.L.myo_last_error().decode()}")

    @property
    def version(self) -> str:
        return self.L.myo_version().decode()

    @property
    def build_id(self) -> str:
        """hash of the native sources this library was compiled from (build.py:source_id), "unknown" for ad-hoc builds"""
        v = self.version
        return v.rsplit(" build ", 1)[1] if " build " in v else "unknown"

    @property
    def is_emulation(self) -> bool:
        return "MYO_EMU" in self.version


_LIB: Optional[NativeLib] = None


def load(path: Optional[str] = None) -> NativeLib:
    """Load the HIP library (or an explicitly named one — tests only)."""
    global _LIB
    if path is not None:
        return NativeLib(path)
    if _LIB is None:
        _LIB = NativeLib(LIB_PATH)
    return _LIB


EXPORTED_SYMBOLS = [
    "myo_model_from_blob", "myo_model_load_mjb", "myo_model_destroy", "myo_model_size", "myo_batch_create",
    "myo_batch_destroy", "myo_batch_set_step_generation", "myo_batch_health", "myo_debug_wave_slots", "myo_batch_num_envs", "myo_batch_obs_dim", "myo_batch_lds_bytes",
    "myo_batch_reset", "myo_batch_step", "myo_batch_step_inner", "myo_batch_step_inner_idx", "myo_batch_copy_envs", "myo_batch_physics_step", "myo_batch_get_state",
    "myo_batch_set_state", "myo_batch_warmstart", "myo_batch_set_bad_state_buffer", "myo_batch_set_task", "myo_batch_get_task", "myo_batch_set_object_group", "myo_batch_object_friction", "myo_batch_bind_constants", "myo_batch_tune_wrap_order", "myo_batch_forward_dump",
    "myo_batch_dump_size", "myo_batch_dump_offset", "myo_batch_kernel_ms",
    "myo_batch_enable_timing", "myo_ppo_loss_grad", "myo_ppo_gather", "myo_bias_relu_bf16", "myo_rollout_policy_input", "myo_rollout_sample",
    "myo_vecnorm_step", "myo_rollout_sample_sde", "myo_vecnorm_batch_moments", "myo_vecnorm_finish", "myo_rollout_advance", "myo_gae", "myo_lstm_cell_fwd", "myo_lstm_cell_bwd", "myo_lstm_step_supported", "myo_lstm_step_fwd", "myo_lstm_step_bwd", "myo_lstm_seq_supported", "myo_lstm_seq_fwd", "myo_lstm_seq_bwd", "myo_splitk_reduce", "myo_splitk_reduce2", "myo_relu_bwd_colsum_bf16", "myo_adam_clip_step", "myo_ppo_mlp_workspace_bytes", "myo_ppo_mlp_step", "myo_ppo_mlp_sqnorm_parts", "myo_adam_apply", "myo_ppo_mlp_rollout_workspace_bytes", "myo_ppo_mlp_rollout_refresh", "myo_ppo_mlp_rollout", "myo_last_error", "myo_version",
]

# ... of which the lane-serial emulation build (tests/emu/libmyobatch_emu.so: csrc/myobatch_emu.cpp + csrc/emu_host.h, test tooling) has the env
# path; the PPO-side kernels exist in the product library only
ENV_PATH_SYMBOLS = [
    "myo_batch_bind_constants", "myo_batch_copy_envs", "myo_batch_create", "myo_batch_destroy", "myo_batch_dump_offset", "myo_batch_dump_size",
    "myo_batch_enable_timing", "myo_batch_forward_dump", "myo_batch_get_state", "myo_batch_get_task", "myo_batch_health", "myo_batch_kernel_ms",
    "myo_batch_lds_bytes", "myo_batch_num_envs", "myo_batch_object_friction", "myo_batch_obs_dim", "myo_batch_physics_step", "myo_batch_reset",
    "myo_batch_set_bad_state_buffer", "myo_batch_set_object_group", "myo_batch_set_state", "myo_batch_set_step_generation", "myo_batch_set_task",
    "myo_batch_step", "myo_batch_step_inner", "myo_batch_step_inner_idx", "myo_batch_tune_wrap_order", "myo_batch_warmstart", "myo_debug_wave_slots",
    "myo_last_error", "myo_model_destroy", "myo_model_from_blob", "myo_model_load_mjb", "myo_model_size", "myo_version",
]


class Model:
    """Host handle of a compiled model (myo_model*)."""

    def __init__(self, compiled, lib: Optional[NativeLib] = None):
        self.lib = lib or load()
        self.compiled = compiled
        blob = compiled.to_blob()
        self._buf = C.create_string_buffer(blob, len(blob))
        h = C.c_void_p()
        self.lib.check(self.lib.L.myo_model_from_blob(self._buf, len(blob), C.byref(h)))
        self.h = h

    @classmethod
    def from_mjb(cls, path: str, lib: Optional[NativeLib] = None, integrator: Optional[int] = None, unsupported_contacts: str = "error",
                 allow_other_solver: bool = False):
        """The C route: libmyobatch reads the .mjb itself (myo_model_load_mjb)."""
        self = cls.__new__(cls)
        self.lib, self.compiled, self._buf = lib or load(), None, None
        h = C.c_void_p()
        self.lib.check(self.lib.L.myo_model_load_mjb(os.fsencode(path), -1 if integrator is None else int(integrator),
                                                     {"error": 0, "drop": 1}[unsupported_contacts] | (2 if allow_other_solver else 0), C.byref(h)))
        self.h = h
        return self

    def size(self, name: str) -> int:
        return self.lib.L.myo_model_size(self.h, name.encode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.L.myo_model_destroy(self.h)
            self.h = None


def _ptr(x):
    """Device/host pointer of a torch tensor, numpy array, int or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data
    return x.data_ptr()  # torch.Tensor


class Batch:
    """Thin handle of myo_batch*; every array argument is a pointer provider (see _ptr)."""

    def __init__(self, model: Model, cfg: Optional[TaskCfg], n_envs: int, device: int = 0, seed: int = 0,
                 dtype: int = MYO_F32):
        self.lib, self.model = model.lib, model
        h = C.c_void_p()
        self.lib.check(self.lib.L.myo_batch_create(model.h, C.byref(cfg) if cfg is not None else None,
                                                   n_envs, device, seed, dtype, C.byref(h)))
        self.h = h
        self.n = n_envs
        self.dtype = dtype
        self.obs_dim = self.lib.L.myo_batch_obs_dim(h)

    def reset(self, mask=None, obs=None, stream=None):
        self.lib.check(self.lib.L.myo_batch_reset(self.h, _ptr(mask), _ptr(obs), stream))

    def step(self, act, obs, rew, done, trunc=None, term_obs=None, comps=None, ep_info=None, stream=None):
        self.lib.check(self.lib.L.myo_batch_step(self.h, _ptr(act), _ptr(obs), _ptr(rew), _ptr(done), _ptr(trunc),
                                                 _ptr(term_obs), _ptr(comps), _ptr(ep_info), stream))

    def step_inner(self, mask, act, obs, done=None, stream=None):
        self.lib.check(self.lib.L.myo_batch_step_inner(self.h, _ptr(mask), _ptr(act), _ptr(obs), _ptr(done), stream))

    def step_inner_idx(self, idx, act, obs, done=None, stream=None):
        """compact form of step_inner: idx int32[n] (-1 = empty slot), act [n, nu], obs [n, obs_dim], done uint8[n] or None"""
        self.lib.check(self.lib.L.myo_batch_step_inner_idx(self.h, _ptr(idx), int(idx.shape[0]), _ptr(act), _ptr(obs), _ptr(done), stream))

    def copy_envs_from(self, src: "Batch", dst_idx, src_idx, stream=None):
        """env records src_idx of `src` -> envs dst_idx of this batch (int32 device arrays of equal length)"""
        self.lib.check(self.lib.L.myo_batch_copy_envs(self.h, _ptr(dst_idx), src.h, _ptr(src_idx), int(dst_idx.shape[0]), stream))

    def physics_step(self, ctrl, nsub, stream=None):
        self.lib.check(self.lib.L.myo_batch_physics_step(self.h, _ptr(ctrl), nsub, stream))

    def get_state(self, qpos=None, qvel=None, act=None, time=None, stream=None):
        self.lib.check(self.lib.L.myo_batch_get_state(self.h, _ptr(qpos), _ptr(qvel), _ptr(act), _ptr(time), stream))

    def set_state(self, qpos=None, qvel=None, act=None, time=None, stream=None):
        self.lib.check(self.lib.L.myo_batch_set_state(self.h, _ptr(qpos), _ptr(qvel), _ptr(act), _ptr(time), stream))

    def health(self) -> dict:
        """{"protocol_errors": k_step workgroups that met a hand-off state of another generation, "contact_overflows": substeps that
        dropped contacts beyond the scratch's capacity, "limit_row_overflows": substeps that dropped limit / friction-loss rows,
        "contact_slots_wanted": the largest number of contact slots such a substep asked for (0 when none overflowed)}; all 0
        in a healthy batch.  Synchronises the device."""
        out = (C.c_int32 * 4)()
        self.lib.check(self.lib.L.myo_batch_health(self.h, out))
        return {"protocol_errors": int(out[0]), "contact_overflows": int(out[1]), "limit_row_overflows": int(out[2]), "contact_slots_wanted": int(out[3])}

    def set_bad_state_buffer(self, buf):
        """uint8[N] device buffer (kept alive by the caller) that every step() fills with 1 for envs reset after a
        numerical blow-up; None to stop reporting."""
        self._bad_buf = buf
        self.lib.check(self.lib.L.myo_batch_set_bad_state_buffer(self.h, _ptr(buf)))

    def warmstart(self, get=None, set=None, stream=None):
        """Read (``get``) and / or overwrite (``set``) qacc_warmstart, double[N, nv]."""
        self.lib.check(self.lib.L.myo_batch_warmstart(self.h, _ptr(get), _ptr(set), stream))

    def set_task(self, task_i=None, task_d=None, ball_d=None, stream=None):
        self.lib.check(self.lib.L.myo_batch_set_task(self.h, _ptr(task_i), _ptr(task_d), _ptr(ball_d), stream))

    def set_object_group(self, gid0: int, gidn: int):
        self.lib.check(self.lib.L.myo_batch_set_object_group(self.h, gid0, gidn))

    def object_friction(self, set_fric=None, get_fric=None, stream=None):
        """friction triples of the object group's geoms, float64 [N, ngeom_group, 3] (set, then get; either may be None)"""
        self.lib.check(self.lib.L.myo_batch_object_friction(self.h, _ptr(set_fric), _ptr(get_fric), stream))

    def bind_constants(self, stream=None):
        """Make this batch the one whose model / task sit in __constant__ memory (needed before replaying a
        captured graph that contains this batch's launches, if other batches may have launched since)."""
        self.lib.check(self.lib.L.myo_batch_bind_constants(self.h, stream))

    def tune_wrap_order(self, stream=None):
        """Re-sort the tendon stage's geom wraps by how often they engage in the envs' present states (runs by itself after a
        reset of all envs; two small kernels on the stream, no result depends on the order: include/myobatch.h)."""
        self.lib.check(self.lib.L.myo_batch_tune_wrap_order(self.h, stream))

    def get_task(self, task_i=None, task_d=None, ball_d=None, stream=None):
        self.lib.check(self.lib.L.myo_batch_get_task(self.h, _ptr(task_i), _ptr(task_d), _ptr(ball_d), stream))

    def forward_dump(self, ctrl, out, stream=None):
        self.lib.check(self.lib.L.myo_batch_forward_dump(self.h, _ptr(ctrl), _ptr(out), stream))

    @property
    def dump_size(self) -> int:
        return self.lib.L.myo_batch_dump_size(self.h)

    def dump_offset(self, name: str) -> int:
        return self.lib.L.myo_batch_dump_offset(self.h, name.encode())

    @property
    def lds_bytes(self) -> int:
        return self.lib.L.myo_batch_lds_bytes(self.h)

    def enable_timing(self, on=True):
        self.lib.check(self.lib.L.myo_batch_enable_timing(self.h, int(on)))

    def kernel_ms(self) -> float:
        return self.lib.L.myo_batch_kernel_ms(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.L.myo_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
