"""ctypes wrapper around oracle/libmyo_oracle.so — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(parity status: see oracle/myo_oracle.h — statics and task layer pinned, stepping unpinned).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Builds libmyo_oracle.so and libmyo_oracle_flops.so (the same source with the flop counter compiled in);
    returns the one this process uses: the counting build iff MYO_ORACLE_FLOPS=1 is set in the environment."""
    name = "libmyo_oracle_flops.so" if os.environ.get("MYO_ORACLE_FLOPS") == "1" else "libmyo_oracle.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "myo_oracle.c")
    stale = lambda f: not os.path.exists(f) or os.path.getmtime(f) < os.path.getmtime(src)
    if force or stale(so) or stale(os.path.join(_HERE, "libmyo_oracle.so")) or stale(os.path.join(_HERE, "libmyo_oracle_flops.so")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_model_from_blob.restype = C.c_void_p
        L.orc_model_from_blob.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_int]
        L.orc_model_free.argtypes = [C.c_void_p]
        L.orc_model_int.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_data_new.restype = C.c_void_p
        L.orc_data_new.argtypes = [C.c_void_p]
        L.orc_data_free.argtypes = [C.c_void_p]
        L.orc_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_ptr.restype = C.POINTER(C.c_double)
        L.orc_ptr.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_count.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_get_int.argtypes = [C.c_void_p, C.c_char_p]
        for f in ("orc_fwd_position", "orc_forward", "orc_step", "orc_kinematics"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_baoding_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_float), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]
        L.orc_baoding_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.orc_baoding_reward.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double),
                                         C.POINTER(C.c_double)]
        dp = C.POINTER(C.c_double)
        L.orc_euler2quat.argtypes = [dp, dp]
        L.orc_reorient_set_die.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, dp, C.c_double]
        L.orc_reorient_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, dp]
        L.orc_reorient_reward.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_reorient_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), dp, dp]
        L.orc_reorient_reset_dists.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, dp]
        L.orc_set_line_search.argtypes = [C.c_int]
        L.orc_flops.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.orc_flops_stages.restype = C.c_int
        L.orc_flops_enabled.restype = C.c_int
        _LIB = L
    return _LIB


def set_line_search(primal: bool) -> None:
    """Newton line search of the oracle: False = safeguarded Newton on p'(alpha) (default); True = the bracketing structure of MuJoCo
    2.1's PrimalSearch (myo_oracle.c:primal_search).  Process-wide."""
    lib().orc_set_line_search(1 if primal else 0)


FLOP_STAGES = ("kinematics", "com", "tendon", "crb", "collision", "constraint", "velocity", "actuation", "acceleration",
               "newton", "integrate", "task")


def flops(reset=False):
    """Floating-point operations counted by the instrumented oracle since the last reset, per stage (dict).
    Needs the counting build: start the process with MYO_ORACLE_FLOPS=1."""
    if not lib().orc_flops_enabled():
        raise RuntimeError("this process loaded the oracle without the flop counter: set MYO_ORACLE_FLOPS=1 before the first use")
    n = lib().orc_flops_stages()
    out = (C.c_double * n)()
    lib().orc_flops(out, int(reset))
    return dict(zip(FLOP_STAGES, out[:]))


class BaodingCfg(C.Structure):
    _fields_ = [("frame_skip", C.c_int), ("obj1_sid", C.c_int), ("obj2_sid", C.c_int),
                ("target1_sid", C.c_int), ("target2_sid", C.c_int), ("obj1_bid", C.c_int),
                ("obj2_bid", C.c_int), ("obj1_gid", C.c_int), ("obj2_gid", C.c_int),
                ("n_hand", C.c_int), ("drop_th", C.c_double), ("proximity_th", C.c_double),
                ("center_pos", C.c_double * 2), ("w", C.c_double * 7)]


class BaodingState(C.Structure):
    _fields_ = [("which_task", C.c_int), ("counter", C.c_int), ("start_angle", C.c_double * 2),
                ("x_radius", C.c_double), ("y_radius", C.c_double), ("time_period", C.c_double)]


class ReorientCfg(C.Structure):
    _fields_ = [("frame_skip", C.c_int), ("n_hand", C.c_int), ("object_sid", C.c_int), ("goal_sid", C.c_int),
                ("object_bid", C.c_int), ("gid0", C.c_int), ("gidn", C.c_int),
                ("drop_th", C.c_double), ("pos_th", C.c_double), ("rot_th", C.c_double),
                ("goal_obj_offset", C.c_double * 3), ("w", C.c_double * 9)]


class ReorientState(C.Structure):
    _fields_ = [("goal_pos", C.c_double * 3), ("goal_quat", C.c_double * 4), ("pos_dist", C.c_double), ("rot_dist", C.c_double)]


REORIENT_RWD_KEYS = ("pos_dist", "rot_dist", "pos_dist_diff", "rot_dist_diff", "alive", "act_reg", "sparse", "solved", "done", "dense")
REWARD_KEYS = ("pos_dist_1", "pos_dist_2", "act_reg", "alive", "sparse", "solved", "done")


class OracleModel:
    def __init__(self, blob: bytes):
        self._blob = blob
        err = C.create_string_buffer(256)
        buf = C.create_string_buffer(blob, len(blob))
        self.h = lib().orc_model_from_blob(buf, len(blob), err, 256)
        if not self.h:
            raise ValueError(err.value.decode())

    def __getattr__(self, name):
        v = lib().orc_model_int(self.h, name.encode())
        if v < 0:
            raise AttributeError(name)
        return v

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_model_free(self.h)
            self.h = None


class OracleData:
    def __init__(self, model: OracleModel):
        self.model = model
        self.h = lib().orc_data_new(model.h)

    def arr(self, name, shape=None):
        n = lib().orc_count(self.h, name.encode())
        if name == "time":
            n = 1
        if n < 0:
            raise KeyError(name)
        p = lib().orc_ptr(self.h, name.encode())
        a = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n]
        return a.reshape(shape) if shape else a

    def __getattr__(self, name):
        if name in ("ncon", "nefc", "solver_iter", "bad", "nl", "ntl"):
            return lib().orc_get_int(self.h, name.encode())
        try:
            return self.arr(name)
        except KeyError as exc:
            raise AttributeError(name) from exc

    def reset(self):
        lib().orc_reset(self.model.h, self.h)

    def set_ball_params(self, cfg, ball_d):
        """Per-env ball physics as CustomBaodingP2Env.reset writes it into its own model
        (/root/reference/src/envs/baoding.py:559-604): ball_d = mass1, mass2, friction1[3], friction2[3],
        size1, size2.  Only body_mass / geom_friction / geom_size[0] change — the reference does not re-run
        mj_setConst (comment at :562), so body_inertia, body_invweight0 and geom_rbound stay nominal, which is
        what orc_forward then sees.  orc_reset restores the model's values: call this again after a reset."""
        ball_d = np.asarray(ball_d, float)
        self.arr("body_mass")[cfg.obj1_bid], self.arr("body_mass")[cfg.obj2_bid] = ball_d[0], ball_d[1]
        fr = self.arr("geom_friction")
        fr[3 * cfg.obj1_gid:3 * cfg.obj1_gid + 3] = ball_d[2:5]
        fr[3 * cfg.obj2_gid:3 * cfg.obj2_gid + 3] = ball_d[5:8]
        sz = self.arr("geom_size")
        sz[3 * cfg.obj1_gid], sz[3 * cfg.obj2_gid] = ball_d[8], ball_d[9]

    def fwd_position(self):
        lib().orc_fwd_position(self.model.h, self.h)

    def forward(self):
        lib().orc_forward(self.model.h, self.h)

    def step(self):
        lib().orc_step(self.model.h, self.h)

    def kinematics(self):
        lib().orc_kinematics(self.model.h, self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_data_free(self.h)
            self.h = None


def make_cfg(ids: dict, frame_skip=10, drop_th=1.25, proximity_th=0.015, weights=None,
             n_hand=23, center_pos=(-0.0125, -0.07)):
    cfg = BaodingCfg()
    cfg.frame_skip = frame_skip
    for k in ("obj1_sid", "obj2_sid", "target1_sid", "target2_sid", "obj1_bid", "obj2_bid",
              "obj1_gid", "obj2_gid"):
        setattr(cfg, k, int(ids[k]))
    cfg.n_hand = n_hand
    cfg.drop_th, cfg.proximity_th = drop_th, proximity_th
    cfg.center_pos[0], cfg.center_pos[1] = center_pos
    weights = weights or {"pos_dist_1": 5.0, "pos_dist_2": 5.0}
    for i, k in enumerate(REWARD_KEYS):
        cfg.w[i] = float(weights.get(k, 0.0))
    return cfg


def baoding_reward(cfg, obs, na=39):
    obs = np.ascontiguousarray(obs, np.float64)
    out = np.zeros(8)
    lib().orc_baoding_reward(C.byref(cfg), na, obs.ctypes.data_as(C.POINTER(C.c_double)),
                             out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def baoding_step(data: OracleData, cfg, st, action):
    a = np.ascontiguousarray(action, np.float32)
    obs = np.zeros(cfg.n_hand + 24 + data.model.na)
    comps = np.zeros(8)
    lib().orc_baoding_step(data.model.h, data.h, C.byref(cfg), C.byref(st),
                           a.ctypes.data_as(C.POINTER(C.c_float)),
                           obs.ctypes.data_as(C.POINTER(C.c_double)),
                           comps.ctypes.data_as(C.POINTER(C.c_double)))
    return obs, comps


# ---- die-reorient task layer
def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def reorient_set_die(data: "OracleData", cfg: ReorientCfg, friction, del_size: float):
    """The die of one episode (reorient.py:136-147): friction float64 [ngeom_die, 3] (None keeps it), one size delta."""
    fr = None if friction is None else np.ascontiguousarray(friction, np.float64).reshape(-1)
    lib().orc_reorient_set_die(data.model.h, data.h, C.byref(cfg), None if fr is None else _dp(fr), float(del_size))


def reorient_reward(cfg: ReorientCfg, na: int, pos_err, rot_err, act, prev_pos_dist: float, prev_rot_dist: float):
    pe, re, a = (np.ascontiguousarray(x, np.float64) for x in (pos_err, rot_err, act))
    c = np.zeros(10)
    lib().orc_reorient_reward(C.byref(cfg), int(na), _dp(pe), _dp(re), _dp(a), float(prev_pos_dist), float(prev_rot_dist), _dp(c))
    return c


def reorient_reset_dists(data: "OracleData", cfg: ReorientCfg, st: ReorientState):
    obs = np.zeros(2 * cfg.n_hand + 18 + data.model.na)
    lib().orc_reorient_reset_dists(data.model.h, data.h, C.byref(cfg), C.byref(st), _dp(obs))
    return obs


def reorient_step(data: "OracleData", cfg: ReorientCfg, st: ReorientState, action):
    a = np.ascontiguousarray(action, np.float32)
    obs, comps = np.zeros(2 * cfg.n_hand + 18 + data.model.na), np.zeros(10)
    lib().orc_reorient_step(data.model.h, data.h, C.byref(cfg), C.byref(st), a.ctypes.data_as(C.POINTER(C.c_float)), _dp(obs), _dp(comps))
    return obs, comps


def euler2quat(e):
    e, q = np.ascontiguousarray(e, np.float64), np.zeros(4)
    lib().orc_euler2quat(_dp(e), _dp(q))
    return q
