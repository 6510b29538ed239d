"""GPU-resident batched Baoding environment with the stable-baselines3 ``VecEnv`` surface.

Replaces the reference's ``SubprocVecEnv([thunk]*16)`` of ``Monitor(TimeLimit(CustomBaoding*Env))``
(/root/reference/src/main_baoding.py:56-65; env classes /root/reference/src/envs/baoding.py:15-647)
with ONE object whose ``num_envs`` environments live in HBM and step in one kernel launch.

Two call styles:

* SB3 protocol (numpy in / numpy out, host ``infos`` list with ``terminal_observation``,
  ``TimeLimit.truncated``, ``episode`` and the reward components) — what SB3's algorithms,
  ``VecNormalize`` and the reference's callbacks (src/metrics/custom_callbacks.py:34,60,75) expect;
* tensor-native fast path ``step_tensor`` / ``reset_tensor``: device ``torch.Tensor`` in and
  out, no host synchronisation — what this repo's PPO engine uses.
"""
from __future__ import annotations

from dataclasses import dataclass
import time
from typing import Any, List, Optional, Sequence

import numpy as np

from .. import native
from ..model import CompiledModel, compile_model
from .config import make_task_cfg, resolve_kwargs


@dataclass
class Box:
    """Stand-in for gym.spaces.Box (gym is not a dependency)."""
    low: np.ndarray
    high: np.ndarray
    shape: tuple
    dtype: Any = np.float32

    @classmethod
    def uniform(cls, lo, hi, n):
        return cls(np.full(n, lo, np.float32), np.full(n, hi, np.float32), (n,), np.float32)

    def sample(self, rng=np.random):
        return rng.uniform(self.low, self.high).astype(self.dtype)


class BaodingVecEnv:
    """``num_envs`` Baoding environments on one GPU (``device`` index)."""

    metadata = {"render.modes": []}

    def __init__(self, env_name: str, num_envs: int, config: Optional[dict] = None, *, device: int = 0,
                 seed: int = 0, dtype: str = "mixed", model=None, integrator: Optional[str] = None,
                 lib: Optional[native.NativeLib] = None, unsupported_contacts: str = "error"):
        import torch
        config = dict(config or {})
        self.env_name = env_name
        self.config = config
        self.params = self._resolve(env_name, config)
        if model is None:
            model = self._default_model()
        if not isinstance(model, CompiledModel):
            integ = None if integrator is None else {"euler": 0, "rk4": 1}[integrator.lower()]
            model = self._compile(model, integ, unsupported_contacts)
            if model.dropped_pairs:            # an explicit opt-in ("drop"): say what the physics now lacks
                import warnings
                warnings.warn(f"{env_name}: {len(model.dropped_pairs)} colliding geom pair(s) have no narrow phase and were dropped: "
                              f"{model.dropped_pairs[:8]}{' ...' if len(model.dropped_pairs) > 8 else ''}")
        self.compiled = model
        self.lib = lib or native.load()
        self.torch = torch
        self.device = self._select_device(device)
        self._model = native.Model(model, self.lib)
        self._cfg = self._make_cfg(env_name, model, config)
        self.max_episode_steps = int(self._cfg.max_episode_steps)
        self.dtype = {"mixed": native.MYO_MIXED, "f32": native.MYO_MIXED, "f64": native.MYO_F64}[dtype]      # "f32": round 1's name of the mixed stepper
        self.batch = native.Batch(self._model, self._cfg, num_envs, device, seed, self.dtype)
        self.num_envs = num_envs
        self.obs_dim = self.batch.obs_dim
        self.act_dim = self._model.size("nu")
        self.observation_space = Box.uniform(-10.0, 10.0, self.obs_dim)   # SB3 zip `data` [ART]
        self.action_space = Box.uniform(-1.0, 1.0, self.act_dim)
        n, o, d = num_envs, self.obs_dim, self.device
        self._obs = torch.zeros((n, o), dtype=torch.float32, device=d)
        self._rew = torch.zeros(n, dtype=torch.float32, device=d)
        self._done = torch.zeros(n, dtype=torch.uint8, device=d)
        self._trunc = torch.zeros(n, dtype=torch.uint8, device=d)
        self._term = torch.zeros((n, o), dtype=torch.float32, device=d)
        self._comps = torch.zeros((n, native.N_RWD), dtype=torch.float32, device=d)
        self._ep = torch.zeros((n, 2), dtype=torch.float32, device=d)
        self._pending = None
        self._closed = False
        self._t_start = time.time()

    def _select_device(self, device: int):
        """The torch device the batch lives on: a GPU, always (libmyobatch has no CPU execution path)."""
        if not self.torch.cuda.is_available():
            raise native.MyoError("BaodingVecEnv needs a GPU: libmyobatch has no CPU execution path")
        return self.torch.device(f"cuda:{device}")

    # ---------------------------------------------------------------- what a task supplies (ReorientVecEnv overrides these)
    rwd_keys = native.RWD_KEYS

    @staticmethod
    def _resolve(env_name, config):
        return resolve_kwargs(env_name, **config)

    @staticmethod
    def _default_model():
        from ..synth_hand import synthetic_hand
        return synthetic_hand()

    @staticmethod
    def _compile(model, integ, unsupported_contacts="error"):
        return compile_model(model, integrator=integ, unsupported_contacts=unsupported_contacts)

    @staticmethod
    def _make_cfg(env_name, compiled, config):
        return make_task_cfg(env_name, compiled, **config)

    # ---------------------------------------------------------------- tensor-native fast path
    def _stream(self):
        if self.device.type != "cuda":
            return None
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def reset_tensor(self):
        self.batch.reset(None, self._obs, self._stream())
        return self._obs

    def step_tensor(self, actions):
        """actions: float32 device tensor [N, 39].  Returns views of internal buffers
        (obs, rew, done, trunc, term_obs, comps, ep_info) — valid until the next call."""
        t = self.torch
        if actions.dtype != t.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=t.float32).contiguous()
        if tuple(actions.shape) != (self.num_envs, self.act_dim):
            raise ValueError(f"actions must have shape {(self.num_envs, self.act_dim)}")
        self.batch.step(actions, self._obs, self._rew, self._done, self._trunc, self._term, self._comps,
                        self._ep, self._stream())
        return self._obs, self._rew, self._done, self._trunc, self._term, self._comps, self._ep

    @property
    def rwd_dict(self):
        """The reward dictionary of the last step as device tensors [N] (what the reference's envs put into `info`)."""
        return {k: self._comps[:, j] for j, k in enumerate(self.rwd_keys)}

    # ---------------------------------------------------------------- SB3 VecEnv protocol
    def reset(self) -> np.ndarray:
        return self.reset_tensor().cpu().numpy().copy()

    def step_async(self, actions) -> None:
        self._pending = self.torch.as_tensor(np.asarray(actions, np.float32), device=self.device)

    def step_wait(self):
        obs, rew, done, trunc, term, comps, ep = self.step_tensor(self._pending)
        self._pending = None
        obs_h, rew_h = obs.cpu().numpy().copy(), rew.cpu().numpy().copy()
        done_h, trunc_h = done.cpu().numpy().astype(bool), trunc.cpu().numpy().astype(bool)
        comps_h = comps.cpu().numpy()
        infos: List[dict] = []
        term_h = ep_h = None
        if done_h.any():
            term_h, ep_h = term.cpu().numpy(), ep.cpu().numpy()
        for i in range(self.num_envs):
            rwd = {k: float(comps_h[i, j]) for j, k in enumerate(self.rwd_keys)}
            info = {"rwd_dense": rwd["dense"], "rwd_sparse": rwd["sparse"], "solved": bool(rwd["solved"]),
                    "done": bool(rwd["done"]), "rwd_dict": rwd}
            if done_h[i]:
                info["terminal_observation"] = term_h[i].copy()
                info["TimeLimit.truncated"] = bool(trunc_h[i])
                info["episode"] = {"r": float(ep_h[i, 0]), "l": int(ep_h[i, 1]), "t": round(time.time() - self._t_start, 6)}   # Monitor's keys
            infos.append(info)
        return obs_h, rew_h, done_h, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        if not self._closed:
            self.batch.close()
            self._closed = True

    def seed(self, seed: Optional[int] = None):
        return [seed] * self.num_envs

    def get_attr(self, attr_name: str, indices: Optional[Sequence[int]] = None):
        idx = range(self.num_envs) if indices is None else indices
        if attr_name == "which_task":
            t = self.torch.zeros((self.num_envs, 2), dtype=self.torch.int32, device=self.device)
            self.batch.get_task(t, None, None, self._stream())
            w = t[:, 0].cpu().numpy()
            return [int(w[i]) for i in idx]
        if attr_name in self.params:
            return [self.params[attr_name] for _ in idx]
        raise AttributeError(attr_name)

    def set_attr(self, attr_name: str, value, indices=None) -> None:
        raise AttributeError("task parameters are fixed at construction; build a new BaodingVecEnv")

    def env_method(self, method_name: str, *args, indices=None, **kwargs):
        raise AttributeError(method_name)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs

    # ---------------------------------------------------------------- state access (parity tests)
    def get_state(self):
        t = self.torch
        n, m = self.num_envs, self._model
        qp = t.zeros((n, m.size("nq")), dtype=t.float64, device=self.device)
        qv = t.zeros((n, m.size("nv")), dtype=t.float64, device=self.device)
        ac = t.zeros((n, m.size("na")), dtype=t.float64, device=self.device)
        tm = t.zeros(n, dtype=t.float64, device=self.device)
        self.batch.get_state(qp, qv, ac, tm, self._stream())
        return qp, qv, ac, tm

    def set_state(self, qpos=None, qvel=None, act=None, time=None):
        c = lambda x: None if x is None else self.torch.as_tensor(x, dtype=self.torch.float64, device=self.device).contiguous()
        args = [c(qpos), c(qvel), c(act), c(time)]
        self.batch.set_state(*args, self._stream())
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
