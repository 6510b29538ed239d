"""Lowering of the reference's Baoding env kwargs to the flat ``myo_task_cfg`` of the C ABI.

Schema = kwargs of ``CustomBaodingEnv._setup`` (/root/reference/src/envs/baoding.py:210-227) and
``CustomBaodingP2Env._setup`` (:300-324), with the gym-registration defaults of
/root/reference/src/envs/__init__.py:12-23 (P1) and :58-74 (P2) applied first — exactly what
``EnvironmentFactory.create(name, **kwargs)`` -> ``gym.make(id, **kwargs)`` does
(/root/reference/src/envs/environment_factory.py:36-37,44-45).  Every archived curriculum
``config.json`` therefore loads unchanged (tests/test_config.py).
"""
from __future__ import annotations

from typing import Dict

from ..native import (CHOICE_CCW, CHOICE_CW, CHOICE_FIXED, CHOICE_RANDOM, TASK_BAODING_P1,
                      TASK_BAODING_P2, TaskCfg)

REWARD_KEYS = ("pos_dist_1", "pos_dist_2", "act_reg", "alive", "sparse", "solved", "done")

# gym registration (src/envs/__init__.py)
REGISTRATION = {
    "CustomMyoBaodingBallsP1": dict(
        variant="p1", max_episode_steps=200,
        kwargs={"normalize_act": True, "goal_xrange": (0.025, 0.025), "goal_yrange": (0.028, 0.028)}),
    "CustomMyoBaodingBallsP2": dict(
        variant="p2", max_episode_steps=200,
        kwargs={"normalize_act": True, "goal_time_period": (4, 6), "goal_xrange": (0.020, 0.030),
                "goal_yrange": (0.022, 0.032), "obj_size_range": (0.018, 0.024),
                "obj_mass_range": (0.030, 0.300), "obj_friction_change": (0.2, 0.001, 0.00002),
                "task_choice": "random"}),
    # src/envs/__init__.py:76-93 (MixtureModelBaoding-v1): the phase-2 env whose reset plays a base policy first
    "MixtureModelBaodingEnv": dict(
        variant="p2", max_episode_steps=200,
        kwargs={"normalize_act": True, "goal_time_period": (4, 6), "goal_xrange": (0.020, 0.030),
                "goal_yrange": (0.022, 0.032), "obj_size_range": (0.018, 0.024),
                "obj_mass_range": (0.030, 0.300), "obj_friction_change": (0.2, 0.001, 0.00002),
                "task_choice": "random"}),
}

# _setup defaults
P1_DEFAULTS = dict(
    frame_skip=10, drop_th=1.25, proximity_th=0.015, goal_time_period=(5, 5), goal_xrange=(0.025, 0.025),
    goal_yrange=(0.028, 0.028),
    weighted_reward_keys={"pos_dist_1": 5.0, "pos_dist_2": 5.0, "alive": 0.0, "act_reg": 0.0},
    task=None, enable_rsi=False, noise_palm=0, noise_fingers=0, noise_balls=0, rsi_probability=1)
P2_DEFAULTS = dict(
    frame_skip=10, drop_th=1.25, proximity_th=0.015, goal_time_period=(5, 5), goal_xrange=(0.025, 0.025),
    goal_yrange=(0.028, 0.028), obj_size_range=(0.018, 0.024), obj_mass_range=(0.030, 0.300),
    obj_friction_change=(0.2, 0.001, 0.00002), task_choice="fixed",
    # BaodingEnvV1.DEFAULT_RWD_KEYS_AND_WEIGHTS [3P-RECALL MyoSuite 1.2.3]
    weighted_reward_keys={"pos_dist_1": 5.0, "pos_dist_2": 5.0},
    enable_rsi=False, rsi_probability=1, balls_overlap=False, overlap_probability=0, limit_init_angle=False,
    beta_init_angle=None, beta_ball_size=None, beta_ball_mass=None, noise_fingers=0)
IGNORED = {"normalize_act", "model_path", "obs_keys", "seed"}


def task_ids(compiled) -> Dict[str, int]:
    """Names the reference resolves with site/body/geom_name2id (baoding.py:264-269,371-378)."""
    return dict(
        obj1_sid=compiled.name2id("site", "ball1_site"), obj2_sid=compiled.name2id("site", "ball2_site"),
        target1_sid=compiled.name2id("site", "target1_site"), target2_sid=compiled.name2id("site", "target2_site"),
        obj1_bid=compiled.name2id("body", "ball1"), obj2_bid=compiled.name2id("body", "ball2"),
        obj1_gid=compiled.name2id("geom", "ball1"), obj2_gid=compiled.name2id("geom", "ball2"))


def resolve_kwargs(env_name: str, **kwargs) -> dict:
    if env_name not in REGISTRATION:
        raise ValueError("Environment name not recognized:", env_name)
    reg = REGISTRATION[env_name]
    defaults = P1_DEFAULTS if reg["variant"] == "p1" else P2_DEFAULTS
    merged = dict(defaults)
    merged.update(reg["kwargs"])
    horizon = kwargs.pop("max_episode_steps", None)     # gym.make(id, max_episode_steps=...) overrides the TimeLimit
    for k, v in kwargs.items():
        if k not in merged and k not in IGNORED:
            raise TypeError(f"{env_name}: unexpected keyword argument {k!r}")
        merged[k] = v
    merged["variant"] = reg["variant"]
    merged["max_episode_steps"] = reg["max_episode_steps"] if horizon is None else int(horizon)
    return merged


def make_task_cfg(env_name: str, compiled, **kwargs) -> TaskCfg:
    p = resolve_kwargs(env_name, **kwargs)
    c = TaskCfg()
    p1 = p["variant"] == "p1"
    c.kind = TASK_BAODING_P1 if p1 else TASK_BAODING_P2
    c.frame_skip = int(p["frame_skip"])
    c.max_episode_steps = int(p["max_episode_steps"])
    c.n_hand = compiled.size("nq") - 14
    for k, v in task_ids(compiled).items():
        setattr(c, k, v)
    if p1:
        task = p["task"]
        if task is None:
            c.task_choice = CHOICE_FIXED
        elif task in ("cw", "ccw", "random"):
            c.task_choice = {"cw": CHOICE_CW, "ccw": CHOICE_CCW, "random": CHOICE_RANDOM}[task]
        else:
            raise ValueError("Unknown task for baoding: ", task)
    else:
        c.task_choice = CHOICE_RANDOM if p["task_choice"] == "random" else CHOICE_FIXED
    c.enable_rsi = int(bool(p["enable_rsi"]))
    c.rsi_probability = float(p["rsi_probability"])
    c.drop_th, c.proximity_th = float(p["drop_th"]), float(p["proximity_th"])
    c.center_pos[0], c.center_pos[1] = -0.0125, -0.07          # baoding.py:250,357
    w = p["weighted_reward_keys"]
    for k in w:
        if k not in REWARD_KEYS:
            raise KeyError(f"unknown reward key {k!r}")
    for i, k in enumerate(REWARD_KEYS):
        c.weights[i] = float(w.get(k, 0.0))
    for name in ("goal_time_period", "goal_xrange", "goal_yrange"):
        arr = getattr(c, name)
        arr[0], arr[1] = float(p[name][0]), float(p[name][1])
    c.noise_fingers = float(p["noise_fingers"])
    if p1:
        c.noise_palm, c.noise_balls = float(p["noise_palm"]), float(p["noise_balls"])
        for v in (c.noise_palm, c.noise_fingers):
            assert 0 <= v <= 1, "Noise must be between 0 and 1"
    else:
        assert 0 <= c.noise_fingers <= 1, "Noise parameter must be between 0 and 1"
        c.balls_overlap = int(bool(p["balls_overlap"]))
        c.overlap_probability = float(p["overlap_probability"])
        lim = p["limit_init_angle"]
        c.limit_init_angle_on = int(bool(lim))
        c.limit_init_angle = float(lim) if lim else 0.0
        for name in ("beta_init_angle", "beta_ball_size", "beta_ball_mass"):
            v = p[name]
            setattr(c, name + "_on", int(bool(v)))
            if v:
                arr = getattr(c, name)
                arr[0], arr[1] = float(v[0]), float(v[1])
        for name in ("obj_size_range", "obj_mass_range"):
            arr = getattr(c, name)
            arr[0], arr[1] = float(p[name][0]), float(p[name][1])
        for i in range(3):
            c.obj_friction_change[i] = float(p["obj_friction_change"][i])
    c.init_qpos0 = -1.57                                         # baoding.py:283,401
    return c
