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
    if use_sde or sd["log_std"].dim() != 1:
        raise NotImplementedError("gSDE checkpoints (log_std matrix) are not supported yet")
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
                            enable_critic_lstm="lstm_critic.weight_hh_l0" in sd or hidden is None)
    pol.load_state_dict(sd, strict=True)
    return pol


def load_policy(path: str) -> Tuple[ActorCriticPolicy, Dict[str, Any]]:
    data, sd, _ = read_zip(path)
    return policy_from_state_dict(sd, bool(data.get("use_sde", False))), data


def save_policy(path: str, policy: ActorCriticPolicy, data: Optional[dict] = None) -> None:
    """Write policy weights + plain hyper-parameters in the same member layout (``data`` carries
    only JSON-plain values; SB3's pickled class objects are not reproduced)."""
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("data", json.dumps({k: v for k, v in (data or {}).items() if isinstance(v, (int, float, bool, str, type(None)))}))
        buf = io.BytesIO()
        torch.save({k: v.detach().cpu() for k, v in policy.state_dict().items()}, buf)
        z.writestr("policy.pth", buf.getvalue())
        z.writestr("_stable_baselines3_version", "myochallenge_amd")
