"""``MyoTrainer`` — same fields and methods as /root/reference/src/train/trainer.py:30-75, on the batched engine.

The reference wraps sb3-contrib's ``RecurrentPPO``: ``_init_agent`` either loads a model zip
(``RecurrentPPO.load(path, env=envs, tensorboard_log=log_dir, custom_objects=model_config)``) or builds a
new ``RecurrentPPO("MlpLstmPolicy", envs, verbose=2, tensorboard_log=log_dir, **model_config)``; ``train``
calls ``agent.learn(total_timesteps, callback=callbacks, reset_num_timesteps=True)``; ``save`` writes
``final_model.pkl`` (a model zip, despite the suffix) and ``final_env.pkl``.  ``model_config`` keys are SB3's
constructor keywords; callables (``lambda _: 5e-05``) are evaluated at progress 1.0 like SB3 schedules.
One addition: ``model_config["policy"] = "MlpPolicy"`` selects the MLP actor-critic (the hipGraph fast
path); the default stays ``"MlpLstmPolicy"``.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Callable, List

from ..rl.policy import ActorCriticPolicy
from ..rl.ppo import PPO, PPOConfig

_PPO_KEYS = ("n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "ent_coef", "vf_coef", "max_grad_norm",
             "normalize_advantage")
_IGNORED = ("verbose", "tensorboard_log", "device", "lr_schedule", "clip_range_vf", "target_kl", "sde_sample_freq",
            "create_eval_env", "_init_setup_model")


def _const(v):
    """SB3 accepts floats or schedules f(progress_remaining); the reference only ever passes constants."""
    return float(v(1.0)) if callable(v) else float(v)


@dataclass
class MyoTrainer:
    envs: object                      # VecNormalize over a batched env
    env_config: dict
    load_model_path: str
    log_dir: str
    model_config: dict = field(default_factory=dict)
    callbacks: List[Callable] = field(default_factory=list)
    timesteps: int = 10_000_000

    def __post_init__(self):
        os.makedirs(self.log_dir, exist_ok=True)
        self.dump_env_config(path=self.log_dir)
        self.agent = self._init_agent()

    def dump_env_config(self, path: str) -> None:
        with open(os.path.join(path, "env_config.json"), "w", encoding="utf8") as f:
            json.dump(self.env_config, f)

    # -- agent construction
    def _ppo_config(self, base: dict) -> PPOConfig:
        mc = self.model_config
        kw = {k: base[k] for k in _PPO_KEYS if k in base}
        kw.update({k: mc[k] for k in _PPO_KEYS if k in mc})
        lr = mc.get("learning_rate", mc.get("lr_schedule", base.get("learning_rate", 3e-4)))
        kw["learning_rate"] = _const(lr)
        kw["clip_range"] = _const(mc.get("clip_range", base.get("clip_range", 0.2)))
        unknown = [k for k in mc if k not in _PPO_KEYS + _IGNORED + ("learning_rate", "clip_range", "policy", "policy_kwargs",
                                                                    "seed", "use_sde")]
        if unknown:
            raise TypeError(f"unexpected model_config keys {unknown}")
        return PPOConfig(**kw)

    def _init_agent(self) -> PPO:
        mc = self.model_config
        env = self.envs
        if self.load_model_path is not None:
            from ..rl.sb3_zip import load_policy
            from ..rl.sb3_zip import read_zip
            policy, data = load_policy(self.load_model_path)      # RecurrentPPO.load(path, env=..., custom_objects=model_config)
            opt_state = read_zip(self.load_model_path)[2]
            if (policy.obs_dim, policy.act_dim) != (env.obs_dim, env.act_dim):
                raise ValueError(f"model expects obs/act {policy.obs_dim}/{policy.act_dim}, env has {env.obs_dim}/{env.act_dim}")
            agent = PPO(env, policy, self._ppo_config(data), seed=int(mc.get("seed") or 0))
            agent.num_timesteps = 0                               # reset_num_timesteps=True
            agent.load_optimizer_state(opt_state)                 # RecurrentPPO.load restores policy.optimizer.pth on every resume
            return agent
        print("\nNo model path provided. Initializing new model.\n")
        pk = dict(mc.get("policy_kwargs") or {})
        arch = pk.get("net_arch", [dict(pi=[64, 64], vf=[64, 64])])
        arch = arch[0] if isinstance(arch, (list, tuple)) and arch and isinstance(arch[0], dict) else arch
        pi, vf = (arch["pi"], arch["vf"]) if isinstance(arch, dict) else (list(arch), list(arch))
        recurrent = mc.get("policy", "MlpLstmPolicy") == "MlpLstmPolicy"
        import torch
        policy = ActorCriticPolicy(env.obs_dim, env.act_dim, pi, vf,
                                   lstm_hidden_size=int(pk.get("lstm_hidden_size", 256)) if recurrent else None,
                                   enable_critic_lstm=bool(pk.get("enable_critic_lstm", True)),
                                   log_std_init=float(pk.get("log_std_init", 0.0)),
                                   activation_fn=pk.get("activation_fn", torch.nn.Tanh),
                                   use_sde=bool(mc.get("use_sde", False)))      # RecurrentPPO(..., use_sde=True, sde_sample_freq=-1)
        return PPO(env, policy, self._ppo_config({}), seed=int(mc.get("seed") or 0))

    # -- the reference's two calls
    def train(self, total_timesteps: int) -> None:
        cbs = list(self.callbacks)
        self.agent.learn(total_timesteps=total_timesteps, callback=(lambda a: [cb(a) for cb in cbs]) if cbs else None)

    def save(self) -> None:
        self.agent.save(os.path.join(self.log_dir, "final_model.pkl"))
        self.envs.save(os.path.join(self.log_dir, "final_env.pkl"))
