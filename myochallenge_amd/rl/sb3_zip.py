"""Read stable-baselines3 / sb3-contrib model zips without SB3 (checkpoint compatibility).

The reference resumes from ``model.zip`` files via ``RecurrentPPO.load(path, env=..., custom_objects=...)``
(/root/reference/src/train/trainer.py:49-56, src/main_eval.py:75-77).  Format [ART, SURVEY.md A.1]:
zip members ``data`` (JSON; non-JSON values base64-cloudpickled under ":serialized:"),
``policy.pth`` (state_dict), ``policy.optimizer.pth``, ``pytorch_variables.pth``,
``_stable_baselines3_version``.  Lambdas (lr_schedule, clip_range) cannot be revived without the
original modules; like the reference we take them from ``custom_objects``.
"""
from __future__ import annotations

import base64
import io
import json
import pickle
import zipfile
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from .policy import ActorCriticPolicy

PLAIN_KEYS = ("n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "ent_coef", "vf_coef",
              "max_grad_norm", "learning_rate", "use_sde", "n_envs", "num_timesteps", "normalize_advantage")


def read_zip(path: str) -> Tuple[Dict[str, Any], Dict[str, torch.Tensor], Optional[dict]]:
    z = zipfile.ZipFile(path)
    data = json.loads(z.read("data"))
    out: Dict[str, Any] = {}
    for k, v in data.items():
        if not isinstance(v, dict):
            out[k] = v
        elif v.get(":type:") == "<class 'numpy.ndarray'>":
            out[k] = pickle.loads(base64.b64decode(v[":serialized:"]))
        elif k == "policy_kwargs":
            out[k] = {kk: vv for kk, vv in v.items() if kk not in (":type:", ":serialized:")}
        elif k in ("observation_space", "action_space"):
            out[k] = {"shape": v.get("shape"), "dtype": v.get("dtype")}
    sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=False)
    opt = None
    if "policy.optimizer.pth" in z.namelist():
        opt = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=False)
    out["_sb3_version"] = z.read("_stable_baselines3_version").decode()
    return out, sd, opt


def policy_from_state_dict(sd: Dict[str, torch.Tensor], use_sde: bool = False) -> ActorCriticPolicy:
    """Infer the architecture from tensor shapes (what policy_kwargs would say)."""
    use_sde = bool(use_sde) or sd["log_std"].dim() == 2        # a gSDE checkpoint's log_std is the [latent_pi, act] matrix
    act_dim = sd["action_net.weight"].shape[0]
    hidden = None
    if "lstm_actor.weight_hh_l0" in sd:
        hidden = sd["lstm_actor.weight_hh_l0"].shape[1]
        obs_dim = sd["lstm_actor.weight_ih_l0"].shape[1]

    def arch(prefix):
        sizes, i = [], 0
        while f"mlp_extractor.{prefix}.{i}.weight" in sd:
            sizes.append(sd[f"mlp_extractor.{prefix}.{i}.weight"].shape[0])
            i += 2
        return sizes
    pi, vf = arch("policy_net"), arch("value_net")
    if hidden is None:
        obs_dim = sd["mlp_extractor.policy_net.0.weight"].shape[1] if pi else sd["action_net.weight"].shape[1]
    pol = ActorCriticPolicy(obs_dim, act_dim, pi, vf, lstm_hidden_size=hidden,
                            enable_critic_lstm="lstm_critic.weight_hh_l0" in sd or hidden is None, use_sde=use_sde)
    pol.load_state_dict(sd, strict=True)
    return pol


def load_policy(path: str) -> Tuple[ActorCriticPolicy, Dict[str, Any]]:
    data, sd, _ = read_zip(path)
    return policy_from_state_dict(sd, bool(data.get("use_sde", False))), data


def _ser(obj, type_name: Optional[str] = None, **plain) -> dict:
    """SB3's ``data_to_json`` entry for a non-JSON value: type string + base64 cloudpickle (+ readable copies)."""
    import cloudpickle
    t = type_name or str(type(obj))
    return {":type:": t, ":serialized:": base64.b64encode(cloudpickle.dumps(obj, protocol=4)).decode(), **plain}


def save_sb3_zip(path: str, policy: ActorCriticPolicy, hyper: Dict[str, Any], *, n_envs: int, num_timesteps: int = 0,
                 n_updates: int = 0, last_obs: Optional[np.ndarray] = None, last_original_obs: Optional[np.ndarray] = None,
                 last_episode_starts: Optional[np.ndarray] = None, optimizer_state: Optional[dict] = None,
                 clip_obs: float = 10.0) -> None:
    """Write a model zip in stable-baselines3 1.6 format (members and ``data`` keys of the reference's
    ``trained_models/**/*.zip``), so that ``RecurrentPPO.load(path, env=..., custom_objects=...)``
    (/root/reference/src/train/trainer.py:49-56, src/main_eval.py:69-77) — or ``PPO.load`` for the MLP
    policy — revives it.  Class-typed entries are written as references to the SB3 / sb3-contrib / gym classes
    (rl/sb3_pickle.py); ``lr_schedule`` / ``clip_range`` are constant functions pickled by value (the
    reference overrides them through ``custom_objects`` anyway)."""
    import collections
    import time
    import cloudpickle  # noqa: F401  (import error here is clearer than inside _ser)
    from .sb3_pickle import instance, stand_ins
    recurrent = policy.recurrent
    specs = [("gym.spaces.box", "Box", "object"),
             ("sb3_contrib.common.recurrent.policies", "RecurrentActorCriticPolicy", "object") if recurrent else
             ("stable_baselines3.common.policies", "ActorCriticPolicy", "object"),
             ("sb3_contrib.common.recurrent.type_aliases", "RNNStates", "namedtuple:pi,vf")]
    O, A = policy.obs_dim, policy.act_dim
    sd = {k: v.detach().cpu().clone() for k, v in policy.state_dict().items()}
    lr = float(hyper.get("learning_rate", 3e-4))
    clip = float(hyper.get("clip_range", 0.2))
    with stand_ins(specs) as C:
        Box, PolicyCls, RNNStates = (C[(m, n)] for m, n, _ in specs)
        box = lambda n, lo, hi: instance(Box, {"dtype": np.dtype("float32"), "shape": (n,), "low": np.full(n, lo, np.float32),
                                              "high": np.full(n, hi, np.float32), "np_random": None})
        if recurrent:
            pk = {"ortho_init": False, "activation_fn": torch.nn.ReLU, "net_arch": [{"pi": list(policy.pi_arch), "vf": list(policy.vf_arch)}],
                  "enable_critic_lstm": bool(policy.enable_critic_lstm), "lstm_hidden_size": int(policy.lstm_hidden_size)}
            H = int(policy.lstm_hidden_size)
            z = lambda: (torch.zeros(1, n_envs, H), torch.zeros(1, n_envs, H))
            lstm_states = RNNStates(z(), z())
        else:
            pk = {"ortho_init": False, "activation_fn": torch.nn.ReLU, "net_arch": [{"pi": list(policy.pi_arch), "vf": list(policy.vf_arch)}]}
            lstm_states = None
        zeros_obs = np.zeros((n_envs, O), np.float32)
        data: Dict[str, Any] = {
            "policy_class": _ser(PolicyCls, "<class 'abc.ABCMeta'>", __module__=PolicyCls.__module__),
            "verbose": int(hyper.get("verbose", 0)),
            "policy_kwargs": _ser(pk, None, **{k: (str(v) if isinstance(v, type) else v) for k, v in pk.items()}),
            "observation_space": _ser(box(O, -clip_obs, clip_obs), "<class 'gym.spaces.box.Box'>", dtype="float32", shape=[O]),
            "action_space": _ser(box(A, -1.0, 1.0), "<class 'gym.spaces.box.Box'>", dtype="float32", shape=[A]),
            "n_envs": int(n_envs), "num_timesteps": int(num_timesteps), "_total_timesteps": int(hyper.get("total_timesteps", num_timesteps)),
            "_num_timesteps_at_start": 0, "seed": hyper.get("seed"), "action_noise": None, "start_time": time.time(),
            "learning_rate": lr, "tensorboard_log": hyper.get("tensorboard_log"),
            "lr_schedule": _ser((lambda v: (lambda _: v))(lr), "<class 'function'>"),
            "_last_obs": _ser(zeros_obs if last_obs is None else np.asarray(last_obs, np.float32), "<class 'numpy.ndarray'>"),
            "_last_episode_starts": _ser(np.ones(n_envs, bool) if last_episode_starts is None else np.asarray(last_episode_starts, bool),
                                         "<class 'numpy.ndarray'>"),
            "_last_original_obs": _ser(zeros_obs.astype(np.float64) if last_original_obs is None else np.asarray(last_original_obs, np.float64),
                                       "<class 'numpy.ndarray'>"),
            "_episode_num": 0, "use_sde": bool(getattr(policy, "use_sde", False)), "sde_sample_freq": -1, "_current_progress_remaining": 0.0,
            "ep_info_buffer": _ser(collections.deque(maxlen=100), "<class 'collections.deque'>"),
            "ep_success_buffer": _ser(collections.deque(maxlen=100), "<class 'collections.deque'>"),
            "_n_updates": int(n_updates), "n_steps": int(hyper.get("n_steps", 256)), "gamma": float(hyper.get("gamma", 0.99)),
            "gae_lambda": float(hyper.get("gae_lambda", 0.95)), "ent_coef": float(hyper.get("ent_coef", 0.0)),
            "vf_coef": float(hyper.get("vf_coef", 0.5)), "max_grad_norm": float(hyper.get("max_grad_norm", 0.5)),
            "batch_size": int(hyper.get("batch_size", 64)), "n_epochs": int(hyper.get("n_epochs", 10)),
            "clip_range": _ser((lambda v: (lambda _: v))(clip), "<class 'function'>"), "clip_range_vf": None,
            "normalize_advantage": bool(hyper.get("normalize_advantage", True)), "target_kl": None,
        }
        if recurrent:
            data["_last_lstm_states"] = _ser(lstm_states, "<class 'sb3_contrib.common.recurrent.type_aliases.RNNStates'>")
        data_json = json.dumps(data, indent=4)
    names = list(sd.keys())
    if optimizer_state is None:        # Adam state of a freshly created optimiser
        optimizer_state = {"state": {}, "param_groups": [{"lr": lr, "betas": (0.9, 0.999), "eps": 1e-5, "weight_decay": 0,
                                                          "amsgrad": False, "maximize": False, "foreach": None,
                                                          "capturable": False, "params": list(range(len(names)))}]}

    def blob(obj) -> bytes:
        buf = io.BytesIO()
        torch.save(obj, buf)
        return buf.getvalue()
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("data", data_json)
        z.writestr("pytorch_variables.pth", blob({}))
        z.writestr("policy.pth", blob(sd))
        z.writestr("policy.optimizer.pth", blob(optimizer_state))
        z.writestr("_stable_baselines3_version", "1.6.0")
        z.writestr("system_info.txt", "written by myochallenge_amd (stable-baselines3 1.6 zip layout)\n")


def save_policy(path: str, policy: ActorCriticPolicy, data: Optional[dict] = None) -> None:
    """Write policy weights + plain hyper-parameters in the same member layout (``data`` carries
    only JSON-plain values; SB3's pickled class objects are not reproduced)."""
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("data", json.dumps({k: v for k, v in (data or {}).items() if isinstance(v, (int, float, bool, str, type(None)))}))
        buf = io.BytesIO()
        torch.save({k: v.detach().cpu() for k, v in policy.state_dict().items()}, buf)
        z.writestr("policy.pth", buf.getvalue())
        z.writestr("_stable_baselines3_version", "myochallenge_amd")
