#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the reference checkout.

Runs ONLY where /root/reference is mounted (the build container).  Nothing from the
reference travels: the outputs are data (inputs + expected outputs, decoded artifacts,
the three small .mjb model data files).  Recipes follow SURVEY.md Appendix A.

  * task layer: /root/reference/src/envs/baoding.py is imported with stub base classes
    (myosuite / sb3 / gym are absent) and its reward + reset functions are called unbound
    on fake ``self`` objects -> reward_goldens.npz, reset_logic_goldens.json
  * artifacts: SB3 zip, VecNormalize pickles, MJB models -> npz/json
"""
import base64
import collections
import enum
import glob
import importlib.util
import io
import json
import os
import pickle
import random
import shutil
import sys
import types
import zipfile

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


# --------------------------------------------------------------------------- stubs
class Task(enum.Enum):  # MyoSuite 1.2.3 baoding_v1.Task [3P-RECALL]
    HOLD = 0
    BAODING_CW = 1
    BAODING_CCW = 2


def import_reference_baoding():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class BaseV0:
        @staticmethod
        def _setup(self, **kw):
            self._base_setup_kwargs = kw

    class BaodingEnvV1(BaseV0):
        DEFAULT_OBS_KEYS = ["hand_pos", "object1_pos", "object1_velp", "object2_pos",
                            "object2_velp", "target1_pos", "target2_pos", "target1_err",
                            "target2_err"]
        DEFAULT_RWD_KEYS_AND_WEIGHTS = {"pos_dist_1": 5.0, "pos_dist_2": 5.0}

    for n in ["myosuite", "myosuite.envs", "myosuite.envs.myo", "myosuite.envs.myo.myochallenge",
              "stable_baselines3", "stable_baselines3.common", "stable_baselines3.common.vec_env",
              "envs"]:
        mod(n)
    mod("myosuite.envs.myo.base_v0", BaseV0=BaseV0)
    mod("myosuite.envs.myo.myochallenge.baoding_v1", WHICH_TASK=Task.BAODING_CCW,
        BaodingEnvV1=BaodingEnvV1, Task=Task)
    mod("sb3_contrib", RecurrentPPO=object)
    sys.modules["stable_baselines3.common.vec_env"].VecNormalize = object
    mod("stable_baselines3.common.vec_env.dummy_vec_env", DummyVecEnv=object)
    mod("envs.environment_factory", EnvironmentFactory=object)
    spec = importlib.util.spec_from_file_location("ref_baoding", f"{REF}/src/envs/baoding.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# --------------------------------------------------------------------------- artifacts
class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, s):
        self.__dict__.update(s if isinstance(s, dict) else {"_state": s})


class StubUnpickler(pickle.Unpickler):
    PASS = ("numpy", "builtins", "collections", "copyreg", "_codecs")

    def find_class(self, module, name):
        if module.split(".")[0] in self.PASS:
            return super().find_class(module, name)
        return type(name, (_Stub,), {"__module__": module})


def load_vecnormalize(path):
    with open(path, "rb") as fh:
        return StubUnpickler(fh).load()


def vecnorm_state(o):
    return {
        "obs_rms": {"mean": o.obs_rms.mean.tolist(), "var": o.obs_rms.var.tolist(),
                    "count": float(o.obs_rms.count)},
        "ret_rms": {"mean": float(o.ret_rms.mean), "var": float(o.ret_rms.var),
                    "count": float(o.ret_rms.count)},
        "clip_obs": float(o.clip_obs), "clip_reward": float(o.clip_reward),
        "gamma": float(o.gamma), "epsilon": float(o.epsilon),
        "norm_obs": bool(o.norm_obs), "norm_reward": bool(o.norm_reward),
        "training": bool(o.training),
    }


def artifacts():
    import torch
    tm = f"{REF}/trained_models"
    # ---- VecNormalize pickles -> stats + raw obs snapshots
    pkls = sorted(p for p in glob.glob(f"{tm}/**/*.pkl", recursive=True)
                  if "scaler" not in p)
    pkls.append(f"{tm}/phase_1/normalized_env_phase1_final")
    stats, snaps = {}, []
    for p in pkls:
        o = load_vecnormalize(p)
        key = os.path.relpath(p, tm)
        stats[key] = vecnorm_state(o)
        if getattr(o, "old_obs", None) is not None:
            snaps.append(np.asarray(o.old_obs, dtype=np.float64).reshape(-1, 86))
    with open(f"{OUT}/vecnormalize_states.json", "w") as fh:
        json.dump(stats, fh)
    allobs = np.unique(np.concatenate(snaps, 0), axis=0)
    np.savez_compressed(f"{OUT}/obs_snapshots.npz", obs=allobs)
    print("vecnormalize pickles:", len(stats), "unique raw obs:", allobs.shape)
    # one raw pickle travels as a data fixture for the loader test (plain pickle of stats)
    shutil.copy(f"{tm}/phase_1/normalized_env_phase1_final", f"{OUT}/normalized_env_phase1_final.pkl")

    # ---- SB3 zip
    zpath = f"{tm}/phase_1/phase1_final.zip"
    shutil.copy(zpath, f"{OUT}/phase1_final.zip")  # data artifact (weights), 2.7 MB
    z = zipfile.ZipFile(zpath)
    data = json.loads(z.read("data"))
    schema = {"members": z.namelist(), "data_keys": sorted(data.keys()),
              "plain": {k: v for k, v in data.items() if not isinstance(v, dict)},
              "policy_kwargs": {k: v for k, v in data["policy_kwargs"].items()
                                if k != ":serialized:"},
              "version": z.read("_stable_baselines3_version").decode()}
    with open(f"{OUT}/sb3_zip_schema.json", "w") as fh:
        json.dump(schema, fh, indent=1)

    def arr(key):
        return pickle.loads(base64.b64decode(data[key][":serialized:"]))

    last_obs = arr("_last_obs")
    last_orig = arr("_last_original_obs")
    starts = arr("_last_episode_starts")
    np.save(f"{OUT}/reset_obs_golden.npy", last_orig[0])
    assert np.all(last_orig == last_orig[0])
    sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=False)
    # stock-PyTorch forward of the stored state_dict on the stored normalised obs (zero state)
    H = sd["lstm_actor.weight_hh_l0"].shape[1]
    la, lc = torch.nn.LSTM(86, H), torch.nn.LSTM(86, H)
    la.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("lstm_actor")})
    lc.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("lstm_critic")})
    x = torch.from_numpy(last_obs).unsqueeze(0)
    with torch.no_grad():
        ha, (ha_n, ca_n) = la(x)
        hc, (hc_n, cc_n) = lc(x)
        mean = ha[0] @ sd["action_net.weight"].T + sd["action_net.bias"]
        value = hc[0] @ sd["value_net.weight"].T + sd["value_net.bias"]
        std = sd["log_std"].exp()
        logp_mean = (-0.5 * np.log(2 * np.pi) - sd["log_std"]).sum().expand(mean.shape[0])
    np.savez_compressed(
        f"{OUT}/phase1_policy_io.npz", last_obs=last_obs, last_original_obs=last_orig,
        episode_starts=starts, mean=mean.numpy(), value=value.numpy(), std=std.numpy(),
        logp_of_mean=logp_mean.numpy(), h_actor=ha_n[0].numpy(), c_actor=ca_n[0].numpy(),
        h_critic=hc_n[0].numpy(), c_critic=cc_n[0].numpy(),
        **{"sd." + k: v.numpy() for k, v in sd.items()})
    print("policy params:", sum(v.numel() for v in sd.values()))

    # ---- curriculum configs
    cfgs = {}
    for p in sorted(glob.glob(f"{tm}/**/config.json", recursive=True)):
        with open(p) as fh:
            cfgs[os.path.relpath(p, tm)] = json.load(fh)
    with open(f"{OUT}/curriculum_configs.json", "w") as fh:
        json.dump(cfgs, fh, indent=1)
    print("configs:", len(cfgs))
    return allobs, cfgs


def mjb_fixtures():
    from myochallenge_amd.mjb import load_mjb
    for rel, short in [("finger/myo_finger_v0.mjb", "finger"),
                       ("finger/motor_finger_v0.mjb", "motor_finger"),
                       ("basic/myo_load.mjb", "load")]:
        src = f"{REF}/data/myosuite/assets/{rel}"
        shutil.copy(src, f"{OUT}/{os.path.basename(rel)}")  # model DATA file (≤11 KB)
        m = load_mjb(src)
        js = {"sizes": m.sizes, "opt": m.opt, "names": m.names,
              "arrays": {k: v.tolist() for k, v in m.arrays.items()
                         if v.size and v.dtype.kind != "S"}}
        with open(f"{OUT}/mjb_{short}.json", "w") as fh:
            json.dump(js, fh)


# --------------------------------------------------------------------------- task layer
def reward_goldens(ref, allobs, cfgs):
    rng = np.random.RandomState(1234)
    # 669 real snapshots + random obs with balls near / below the drop threshold
    rnd = rng.normal(0, 0.05, size=(256, 86))
    rnd[:, 25] = rng.uniform(1.2, 1.5, 256)
    rnd[:, 31] = rng.uniform(1.2, 1.5, 256)
    rnd[:, 41:47] = rng.normal(0, 0.012, size=(256, 6))
    rnd[:, 47:] = rng.uniform(0, 1, size=(256, 39))
    obs = np.concatenate([allobs, rnd], 0)
    wsets = []
    for c in cfgs.values():
        w = c.get("weighted_reward_keys")
        if w and w not in wsets:
            wsets.append(w)
    wsets.append(dict(ref.CustomBaodingEnv.DEFAULT_RWD_KEYS_AND_WEIGHTS))
    wsets.append({"pos_dist_1": 5.0, "pos_dist_2": 5.0})
    keys = ["pos_dist_1", "pos_dist_2", "act_reg", "alive", "sparse", "solved", "done", "dense"]
    out = {}
    for variant, cls in (("p1", ref.CustomBaodingEnv), ("p2", ref.CustomBaodingP2Env)):
        for th_i, (drop_th, prox) in enumerate(((1.25, 0.015), (1.3, 0.015), (1.3, 0.02))):
            res = np.zeros((len(wsets), obs.shape[0], len(keys)))
            for wi, w in enumerate(wsets):
                for oi, o in enumerate(obs):
                    fake = types.SimpleNamespace(
                        drop_th=drop_th, proximity_th=prox, rwd_keys_wt=w,
                        obs_dict={"act": o[47:86]}, object1_gid=0, object2_gid=1,
                        sim=types.SimpleNamespace(model=types.SimpleNamespace(
                            na=39, geom_rgba=np.zeros((2, 4)))))
                    od = {"target1_err": o[41:44], "target2_err": o[44:47],
                          "object1_pos": o[23:26], "object2_pos": o[29:32]}
                    rd = cls.get_reward_dict(fake, od)
                    res[wi, oi] = [float(rd[k]) for k in keys]
            out[f"{variant}_th{th_i}"] = res
    np.savez_compressed(
        f"{OUT}/reward_goldens.npz", obs=obs, keys=np.array(keys),
        thresholds=np.array([(1.25, 0.015), (1.3, 0.015), (1.3, 0.02)]),
        weight_sets=np.array([json.dumps(w) for w in wsets]), **out)
    print("reward goldens:", obs.shape[0], "obs x", len(wsets), "weight sets")


class Recorder:
    def __init__(self):
        self.calls = []

    def rec(self, name, **kw):
        self.calls.append({"call": name, **{k: (v.tolist() if isinstance(v, np.ndarray) else v)
                                            for k, v in kw.items()}})


def reset_goldens(ref, cfgs):
    """Drive the reference reset() on a fake self and record every externally visible effect.

    Pins T4/T5 of SURVEY.md §8a: which qpos slots are written, with which distributions,
    in which RNG call order, and the probabilities of the RSI / overlap branches."""
    cases = []
    p1_keys = ["task", "enable_rsi", "noise_palm", "noise_fingers", "noise_balls",
               "rsi_probability", "goal_time_period", "goal_xrange", "goal_yrange"]
    uniq = []
    for c in cfgs.values():
        c2 = {k: v for k, v in c.items() if k != "weighted_reward_keys"}
        if c2 not in uniq:
            uniq.append(c2)
    uniq.append({})  # registration defaults
    for ci, cfg in enumerate(uniq):
        is_p2 = any(k in cfg for k in ("task_choice", "obj_size_range", "obj_mass_range",
                                       "balls_overlap", "limit_init_angle", "beta_init_angle"))
        if cfg == {}:
            variants = ("p1", "p2")
        else:
            variants = ("p2",) if is_p2 else ("p1",)
        for variant in variants:
            for seed in range(6):
                rec = Recorder()
                fake_obs = np.random.RandomState(1000 + seed).uniform(-0.5, 0.5, 86)
                init_qpos = np.zeros(37)
                init_qpos[0] = -1.57
                init_qpos[23:30] = [-0.227, -0.511, 1.452, 1, 0, 0, 0]
                init_qpos[30:37] = [-0.256, -0.552, 1.442, 1, 0, 0, 0]

                class Fake:
                    pass
                f = Fake()
                f.np_random = np.random.RandomState(seed)
                np.random.seed(seed + 77)
                random.seed(seed + 99)
                f.dt = 0.02
                f.init_qpos, f.init_qvel = init_qpos, np.zeros(35)
                f.robot = types.SimpleNamespace(
                    reset=lambda qp, qv: rec.rec("robot.reset", qpos=qp.copy(), qvel=qv.copy()))
                f.step = lambda a: rec.rec("step", action=np.asarray(a))
                f.get_obs = lambda: (rec.rec("get_obs"), fake_obs)[1]
                f.set_state = lambda qp, qv: rec.rec("set_state", qpos=qp.copy(), qvel=qv.copy())
                f.create_goal_trajectory = lambda time_step, time_period: (
                    rec.rec("create_goal_trajectory", time_step=time_step,
                            time_period=float(time_period)), np.zeros(4))[1]
                f.sim = types.SimpleNamespace(model=types.SimpleNamespace(
                    body_mass=np.zeros(4), geom_friction=np.tile([1.0, 0.005, 0.0001], (4, 1)),
                    geom_size=np.zeros((4, 3))))
                f.object1_bid, f.object2_bid, f.object1_gid, f.object2_gid = 1, 2, 1, 2
                f.counter = 17
                if variant == "p1":
                    cls = ref.CustomBaodingEnv
                    d = dict(task=None, enable_rsi=False, noise_palm=0, noise_fingers=0,
                             noise_balls=0, rsi_probability=1, goal_time_period=(5, 5),
                             goal_xrange=(0.025, 0.025), goal_yrange=(0.028, 0.028))
                    d.update({k: v for k, v in cfg.items() if k in p1_keys})
                    f.task, f.rsi = d["task"], d["enable_rsi"]
                    for k in ("noise_palm", "noise_fingers", "noise_balls", "rsi_probability",
                              "goal_time_period", "goal_xrange", "goal_yrange"):
                        setattr(f, k, d[k])
                    f.sample_task = lambda: cls.sample_task(f)
                    f._add_noise_to_palm_position = lambda q, n: cls._add_noise_to_palm_position(f, q, n)
                    f._add_noise_to_finger_positions = lambda q, n: cls._add_noise_to_finger_positions(f, q, n)
                else:
                    cls = ref.CustomBaodingP2Env
                    d = dict(task_choice="random", enable_rsi=False, rsi_probability=1,
                             balls_overlap=False, overlap_probability=0, limit_init_angle=False,
                             beta_init_angle=None, beta_ball_size=None, beta_ball_mass=None,
                             noise_fingers=0, goal_time_period=(4, 6), goal_xrange=(0.020, 0.030),
                             goal_yrange=(0.022, 0.032), obj_size_range=(0.018, 0.024),
                             obj_mass_range=(0.030, 0.300),
                             obj_friction_change=(0.2, 0.001, 0.00002))
                    d.update({k: v for k, v in cfg.items() if k in d})
                    f.task_choice, f.rsi = d["task_choice"], d["enable_rsi"]
                    for k in ("rsi_probability", "balls_overlap", "overlap_probability",
                              "limit_init_angle", "beta_init_angle", "beta_ball_size",
                              "beta_ball_mass", "noise_fingers", "goal_time_period",
                              "goal_xrange", "goal_yrange"):
                        setattr(f, k, d[k])
                    f.obj_mass_range = {"low": d["obj_mass_range"][0], "high": d["obj_mass_range"][1]}
                    f.obj_size_range = {"low": d["obj_size_range"][0], "high": d["obj_size_range"][1]}
                    nominal = np.array([1.0, 0.005, 0.0001])
                    f.obj_friction_range = {"low": nominal - d["obj_friction_change"],
                                            "high": nominal + d["obj_friction_change"]}
                    f.which_task = Task.BAODING_CCW
                    f.ball_1_starting_angle = np.pi / 4
                    f.ball_2_starting_angle = np.pi / 4 - np.pi
                    f._add_noise_to_finger_positions = lambda q, n: cls._add_noise_to_finger_positions(f, q, n)
                obs = cls.reset(f)
                case = {
                    "variant": variant, "config_index": ci, "config": d, "seed": seed,
                    "fake_obs": fake_obs.tolist(), "calls": rec.calls,
                    "which_task": int(f.which_task.value),
                    "ball_1_starting_angle": float(f.ball_1_starting_angle),
                    "ball_2_starting_angle": float(f.ball_2_starting_angle),
                    "x_radius": float(f.x_radius), "y_radius": float(f.y_radius),
                    "counter": int(f.counter),
                    "body_mass": f.sim.model.body_mass.tolist(),
                    "geom_friction": f.sim.model.geom_friction.tolist(),
                    "geom_size": f.sim.model.geom_size.tolist(),
                    "returned_obs_is_fake": bool(np.all(obs == fake_obs)),
                }
                cases.append(case)
    with open(f"{OUT}/reset_logic_goldens.json", "w") as fh:
        json.dump(cases, fh)
    print("reset goldens:", len(cases))


def reorient_goldens():
    """Die-reorient reward dictionary: /root/reference/src/envs/reorient.py:12-56 called unbound on fake ``self``
    objects (one env at a time, as the reference does) for 256 random (pos_err, rot_err, act, previous distances)
    and three weight sets -> reorient_reward_goldens.npz.  euler2quat is only imported by the module, not used
    by the reward, so a placeholder is enough."""
    import types as _t
    for n in ["myosuite.envs.env_base", "myosuite.envs.myo.myochallenge.reorient_v0", "myosuite.utils", "myosuite.utils.quat_math"]:
        if n not in sys.modules:
            sys.modules[n] = _t.ModuleType(n)
    sys.modules["myosuite.envs.env_base"].MujocoEnv = object
    base = sys.modules["myosuite.envs.myo.base_v0"].BaseV0

    class ReorientEnvV0(base):
        DEFAULT_OBS_KEYS = ["hand_qpos", "hand_qvel", "obj_pos", "goal_pos", "pos_err", "obj_rot", "goal_rot", "rot_err"]
        DEFAULT_RWD_KEYS_AND_WEIGHTS = {"pos_dist": 100.0, "rot_dist": 1.0}
    sys.modules["myosuite.envs.myo.myochallenge.reorient_v0"].ReorientEnvV0 = ReorientEnvV0
    sys.modules["myosuite.utils.quat_math"].euler2quat = lambda e: None
    spec = importlib.util.spec_from_file_location("ref_reorient", f"{REF}/src/envs/reorient.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.RandomState(7)
    weight_sets = [{"pos_dist": 100.0, "rot_dist": 1.0},
                   {"pos_dist": 1.0, "rot_dist": 1.0, "pos_dist_diff": 100.0, "rot_dist_diff": 10.0, "alive": 1.0, "act_reg": 0.1,
                    "solved": 2.0, "done": -10.0, "sparse": 0.0},
                   {"sparse": 1.0, "solved": 10.0}]
    keys = ["pos_dist", "rot_dist", "pos_dist_diff", "rot_dist_diff", "alive", "act_reg", "sparse", "solved", "done", "dense"]
    n = 256
    pos_err = rng.normal(0, 0.03, (n, 3)) * (rng.rand(n, 1) < 0.85) + rng.normal(0, 0.2, (n, 3)) * (rng.rand(n, 1) < 0.2)
    rot_err = rng.normal(0, 0.6, (n, 3)) * (rng.rand(n, 1) < 0.8)
    pos_err[:8] *= 0.1; rot_err[:8] *= 0.1                     # some solved cases
    act = rng.rand(n, 39)
    prev_p, prev_r = np.abs(rng.normal(0, 0.05, n)), np.abs(rng.normal(0, 1.0, n))
    out = np.zeros((len(weight_sets), n, len(keys)))
    for wi, wts in enumerate(weight_sets):
        for i in range(n):
            f = _t.SimpleNamespace()
            # MyoSuite keeps obs_dict entries as (1, 1, n) arrays, so every reward term is (1, 1)-shaped
            f.obs_dict = {"pos_err": pos_err[i].reshape(1, 1, 3), "rot_err": rot_err[i].reshape(1, 1, 3), "act": act[i].reshape(1, 1, -1)}
            f.pos_dist, f.rot_dist = np.full((1, 1), prev_p[i]), np.full((1, 1), prev_r[i])
            f.sim = _t.SimpleNamespace(model=_t.SimpleNamespace(na=39, site_rgba=np.ones((4, 4))))
            f.drop_th, f.pos_th, f.rot_th = 0.200, 0.025, 0.262
            f.rwd_keys_wt = wts
            f.success_indicator_sid = 1
            rd = m.CustomReorientEnv.get_reward_dict(f, f.obs_dict)
            out[wi, i] = [float(np.asarray(rd[k]).reshape(-1)[0]) for k in keys]
    np.savez(f"{OUT}/reorient_reward_goldens.npz", pos_err=pos_err, rot_err=rot_err, act=act, prev_pos_dist=prev_p, prev_rot_dist=prev_r,
             expected=out, keys=np.array(keys), weights=np.array(json.dumps(weight_sets)), drop_th=0.200, pos_th=0.025, rot_th=0.262)
    print("reorient reward goldens:", out.shape, "solved", int(out[0, :, 7].sum()), "dropped", int(out[0, :, 8].sum()))


def classifier_goldens(allobs):
    """Task classifier of the winning ensemble (src/models/classifier.py:160-174, used by
    src/eval_mixture_of_ensembles.py:139-190): the reference's own TaskClassifier class (imported with the
    stubs above) + its shipped classifier.pt / scaler.pkl evaluated on windows of 13 consecutive archived
    observations [29:47] -> classifier_goldens.npz.  The two artifacts are copied as data fixtures."""
    import torch
    import warnings
    spec = importlib.util.spec_from_file_location("ref_classifier", f"{REF}/src/models/classifier.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)          # needs import_reference_baoding() to have installed the stub modules
    cdir = f"{REF}/trained_models/winning_ensemble/classifier"
    shutil.copy(f"{cdir}/classifier.pt", f"{OUT}/classifier.pt")
    shutil.copy(f"{cdir}/scaler.pkl", f"{OUT}/classifier_scaler.pkl")
    clf = m.TaskClassifier(m.N_OBS_PER_TRIAL)
    clf.load_state_dict(torch.load(f"{cdir}/classifier.pt", map_location="cpu"))
    clf.eval()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scaler = pickle.load(open(f"{cdir}/scaler.pkl", "rb"))
    rows = np.asarray(allobs, np.float64)[:, 29:47]
    rng = np.random.RandomState(0)
    starts = rng.randint(0, len(rows) - 13, size=64)
    X = np.stack([rows[s:s + 13].reshape(-1) for s in starts])           # [64, 234] raw windows
    Xs = scaler.transform(X)
    with torch.no_grad():
        logits = clf(torch.FloatTensor(Xs)).numpy().reshape(-1)
    task = np.round(1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(np.int64)   # update_task()
    np.savez(f"{OUT}/classifier_goldens.npz", windows=X, scaled=Xs, logits=logits, task=task,
             scaler_mean=scaler.mean_, scaler_scale=scaler.scale_)
    print("classifier goldens:", X.shape, "hold fraction", float((task == 0).mean()))


if __name__ == "__main__":
    ref = import_reference_baoding()
    allobs, cfgs = artifacts()
    mjb_fixtures()
    reward_goldens(ref, allobs, cfgs)
    reset_goldens(ref, cfgs)
    classifier_goldens(allobs)
    reorient_goldens()
