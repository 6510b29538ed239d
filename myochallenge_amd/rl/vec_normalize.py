"""Device-resident ``VecNormalize`` with stable-baselines3's state format and update rule.

Replaces ``stable_baselines3.common.vec_env.VecNormalize`` as the reference uses it:
``VecNormalize.load(path, venv)`` (/root/reference/src/main_baoding.py:75), ``.normalize_obs`` and
``.save`` (/root/reference/src/metrics/custom_callbacks.py:34,60), ``envs.save``
(/root/reference/src/train/trainer.py:75), ``.training`` / ``.norm_reward``
(/root/reference/src/main_eval.py:65-67).  Update rule [3P-RECALL SB3 1.6.2, SURVEY.md C.2]:
Chan's parallel mean/variance merge with initial count 1e-4, ``ret = ret*gamma + r``,
clip(obs) ±10, clip(reward) ±10, epsilon 1e-8.  The pickles the reference ships are read
without SB3 through a ``find_class`` remap; statistics live on the device in
float64 so the running moments match the host implementation.
"""
from __future__ import annotations

import pickle


import numpy as np
import torch


class RunningMeanStd:
    def __init__(self, shape=(), device="cpu", epsilon: float = 1e-4):
        # one storage [mean | var | count] (views below): the HIP normaliser kernel updates it in place
        n = int(np.prod(shape)) if shape != () else 1
        self.buf = torch.zeros(2 * n + 1, dtype=torch.float64, device=device)
        self.mean, self.var, self.count = self.buf[:n].view(shape), self.buf[n:2 * n].view(shape), self.buf[2 * n]
        self.var.fill_(1.0)
        self.count.fill_(float(epsilon))           # on the device: no host sync per step

    def load(self, mean, var, count) -> None:
        self.mean.copy_(torch.as_tensor(np.asarray(mean, np.float64)).view(self.mean.shape))
        self.var.copy_(torch.as_tensor(np.asarray(var, np.float64)).view(self.var.shape))
        self.count.fill_(float(count))

    sync = False      # all-reduce the batch moments over torch.distributed ranks (VecNormalize(sync_ranks=True))

    def update(self, x: torch.Tensor) -> None:
        x = x.to(torch.float64)
        if self.sync and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # one VecNormalize over the envs of ALL ranks: additive sums (n, sum x, sum x^2), all-reduced
            n = torch.tensor([float(x.shape[0])], dtype=torch.float64, device=x.device)
            s = torch.cat([n, x.sum(0).reshape(-1), (x * x).sum(0).reshape(-1)])
            torch.distributed.all_reduce(s)
            k = (s.numel() - 1) // 2
            mean = (s[1:1 + k] / s[0]).view(self.mean.shape)
            var = torch.clamp(s[1 + k:] / s[0] - (s[1:1 + k] / s[0]) ** 2, min=0.0).view(self.mean.shape)
            self.update_from_moments(mean, var, s[0])
            return
        if x.is_cuda and x.dim() == 2:
            # column moments as [1,N] x [N,O] products: ATen's multi-block column reduction is not replay-safe
            # inside a hipGraph on this stack (see rl/policy.py:_LinearGemmBias); GEMMs are
            n = x.shape[0]
            ones = torch.ones((1, n), dtype=torch.float64, device=x.device)
            mean = (ones @ x)[0] / n
            d = x - mean
            self.update_from_moments(mean, (ones @ (d * d))[0] / n, n)
            return
        self.update_from_moments(x.mean(0), x.var(0, unbiased=False), x.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count) -> None:
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m2 = self.var * self.count + batch_var * batch_count + delta * delta * self.count * batch_count / tot
        self.mean.copy_(new_mean)          # in place: the tensors keep their addresses (hipGraph replay)
        self.var.copy_(m2 / tot)
        self.count.copy_(tot)

    def state(self):
        return {"mean": self.mean.cpu().numpy().copy(), "var": self.var.cpu().numpy().copy(), "count": float(self.count)}

    def snapshot(self) -> torch.Tensor:
        return self.buf.clone()

    def merge_ranks(self, snap: torch.Tensor) -> None:
        """Per-rollout rank synchronisation (VecNormalize(sync_ranks="rollout")): every rank has updated its own copy from `snap`
        (identical on all ranks) with its own envs' data; afterwards every rank holds merge(snap, data of ALL ranks).  The data a
        rank added is recovered from (snap, state) by inverting Chan's merge — additive sums (n, sum x, sum x^2) — all-reduced once,
        and merged into the snapshot.  One collective per rollout instead of one per env step; the result equals the per-step
        synchronisation up to the order of the floating-point merges."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1):
            return
        n = self.mean.numel()
        m0, v0, c0 = snap[:n], snap[n:2 * n], snap[2 * n]
        m1, v1, c1 = self.mean.reshape(-1), self.var.reshape(-1), self.count
        nd = c1 - c0
        s1 = m1 * c1 - m0 * c0                                              # sum of this rank's samples
        safe = torch.clamp(nd, min=1e-300)
        md = s1 / safe
        m2d = v1 * c1 - v0 * c0 - (md - m0) ** 2 * c0 * nd / c1            # their sum of squared deviations about md
        s = torch.cat([nd.reshape(1), s1, torch.clamp(m2d, min=0.0) + nd * md * md])
        torch.distributed.all_reduce(s)
        tot = s[0]
        self.buf.copy_(snap)
        if float(tot) <= 0.0:
            return
        mean = s[1:1 + n] / tot
        var = torch.clamp(s[1 + n:] / tot - mean * mean, min=0.0)
        self.update_from_moments(mean.view(self.mean.shape), var.view(self.mean.shape), tot)


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, s):
        self.__dict__.update(s if isinstance(s, dict) else {"_state": s})


class _StubUnpickler(pickle.Unpickler):
    _PASS = ("numpy", "builtins", "collections", "copyreg", "_codecs")

    def find_class(self, module, name):
        if module.split(".")[0] in self._PASS:
            return super().find_class(module, name)
        return type(name, (_Stub,), {"__module__": module})


class VecNormalize:
    """Wraps a tensor-native batched env (``reset_tensor`` / ``step_tensor``)."""

    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0,
                 gamma=0.99, epsilon=1e-8, sync_ranks=True):
        """sync_ranks: with torch.distributed initialised, True / "step": the batch moments of every update are all-reduced, so N
        ranks x M envs keep the statistics ONE VecNormalize over N M envs would (the reference wraps all its envs in
        one, /root/reference/src/main_baoding.py:74-75) and every rank holds — and saves — the same normaliser.
        "rollout": every rank updates from its own envs and the ranks' statistics are merged once per rollout
        (begin_rollout / end_rollout, called by PPO): one collective per rollout instead of one per env step, the same
        statistics up to the order of the merges.  False / "none": no exchange."""
        self.venv = venv
        self.num_envs = venv.num_envs if venv is not None else 0
        dev = venv.device if venv is not None else "cpu"
        self.device = dev
        obs_dim = venv.obs_dim if venv is not None else 0
        self.obs_rms = RunningMeanStd((obs_dim,), dev)
        self.ret_rms = RunningMeanStd((), dev)
        self.sync_mode = {True: "step", False: "none", None: "none"}.get(sync_ranks, sync_ranks)
        if self.sync_mode not in ("step", "rollout", "none"):
            raise ValueError(f"sync_ranks must be True / False / 'step' / 'rollout' / 'none', not {sync_ranks!r}")
        self.sync_ranks = self.sync_mode == "step"           # (what the per-update paths test)
        self.obs_rms.sync = self.ret_rms.sync = self.sync_ranks
        self._rollout_snap = None
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        self.returns = torch.zeros(self.num_envs, dtype=torch.float64, device=dev)
        self.old_obs = None
        self.old_reward = None
        if venv is not None:
            self.observation_space, self.action_space = venv.observation_space, venv.action_space
            self.obs_dim, self.act_dim = venv.obs_dim, venv.act_dim

    # -- per-rollout rank synchronisation (sync_ranks="rollout")
    def begin_rollout(self) -> None:
        if self.sync_mode == "rollout" and self.training and self._rollout_snap is None:
            self._rollout_snap = (self.obs_rms.snapshot(), self.ret_rms.snapshot())

    def end_rollout(self) -> None:
        if self._rollout_snap is not None:
            self.obs_rms.merge_ranks(self._rollout_snap[0])
            self.ret_rms.merge_ranks(self._rollout_snap[1])
            self._rollout_snap = None

    # -- normalisation
    def normalize_obs(self, obs):
        is_np = isinstance(obs, np.ndarray)
        o = torch.as_tensor(obs, device=self.device)
        if self.norm_obs:
            o = torch.clamp((o.to(torch.float64) - self.obs_rms.mean) / torch.sqrt(self.obs_rms.var + self.epsilon),
                            -self.clip_obs, self.clip_obs).to(torch.float32)
        return o.cpu().numpy() if is_np else o

    def normalize_reward(self, r):
        if self.norm_reward:
            r = torch.clamp(r.to(torch.float64) / torch.sqrt(self.ret_rms.var + self.epsilon), -self.clip_reward,
                            self.clip_reward).to(torch.float32)
        return r

    def unnormalize_obs(self, obs):
        if not self.norm_obs:
            return obs
        return (obs.to(torch.float64) * torch.sqrt(self.obs_rms.var + self.epsilon) + self.obs_rms.mean).to(torch.float32)

    def get_original_obs(self):
        return self.old_obs

    def get_original_reward(self):
        return self.old_reward

    # -- stepping (tensor-native)
    def reset_tensor(self):
        obs = self.venv.reset_tensor()
        self.old_obs = obs.clone()
        self.returns.zero_()
        if self.training and self.norm_obs:
            self.obs_rms.update(obs)
        return self.normalize_obs(obs)

    def process_step(self, obs, rew, done, trunc, term, comps, ep):
        """Normaliser update + normalisation of one raw env step (pure tensor ops, capturable)."""
        self.old_obs, self.old_reward = obs.clone(), rew.clone()
        if self.training and self.norm_obs:
            self.obs_rms.update(obs)
        if self.training:
            self.returns.mul_(self.gamma).add_(rew.to(torch.float64))
            self.ret_rms.update(self.returns)
        nrew = self.normalize_reward(rew)
        nobs = self.normalize_obs(obs)
        nterm = self.normalize_obs(term)
        self.returns.mul_(1.0 - done.to(torch.float64))
        return nobs, nrew, done, trunc, nterm, comps, ep

    def step_tensor(self, actions):
        return self.process_step(*self.venv.step_tensor(actions))

    # -- SB3 protocol (numpy)
    def reset(self):
        return self.reset_tensor().cpu().numpy()

    def step(self, actions):
        a = torch.as_tensor(np.asarray(actions, np.float32), device=self.device)
        nobs, nrew, done, trunc, nterm, comps, ep = self.step_tensor(a)
        done_h, trunc_h = done.cpu().numpy().astype(bool), trunc.cpu().numpy().astype(bool)
        infos = []
        nterm_h, ep_h = nterm.cpu().numpy(), ep.cpu().numpy()
        for i in range(self.num_envs):
            info = {}
            if done_h[i]:
                info = {"terminal_observation": nterm_h[i].copy(), "TimeLimit.truncated": bool(trunc_h[i]),
                        "episode": {"r": float(ep_h[i, 0]), "l": int(ep_h[i, 1])}}
            infos.append(info)
        return nobs.cpu().numpy(), nrew.cpu().numpy(), done_h, infos

    def close(self):
        if self.venv is not None:
            self.venv.close()

    # -- persistence
    def state_dict(self):
        return {"obs_rms": self.obs_rms.state(), "ret_rms": self.ret_rms.state(), "clip_obs": self.clip_obs,
                "clip_reward": self.clip_reward, "gamma": self.gamma, "epsilon": self.epsilon,
                "norm_obs": self.norm_obs, "norm_reward": self.norm_reward, "training": self.training}

    def save(self, path: str) -> None:
        """``envs.save(path)`` (/root/reference/src/train/trainer.py:75): stable-baselines3's pickle layout."""
        self.save_sb3(path)

    def save_plain(self, path: str) -> None:
        """The statistics as a plain dict (no class references); ``load`` reads both layouts."""
        with open(path, "wb") as fh:
            pickle.dump({"format": "myochallenge_amd.VecNormalize/1", **self.state_dict()}, fh)

    def save_sb3(self, path: str) -> None:
        """Write the pickle ``stable_baselines3.common.vec_env.VecNormalize.load`` reads (what the reference's
        ``envs.save(path)`` produces, /root/reference/src/train/trainer.py:75): a VecNormalize object whose
        state is SB3's ``__getstate__`` (no venv / class_attributes / returns), with ``RunningMeanStd`` and
        gym ``Box`` members.  Works without SB3 / gym installed (see rl/sb3_pickle.py)."""
        from .sb3_pickle import instance, stand_ins
        specs = [("stable_baselines3.common.vec_env.vec_normalize", "VecNormalize", "object"),
                 ("stable_baselines3.common.running_mean_std", "RunningMeanStd", "object"),
                 ("gym.spaces.box", "Box", "object")]
        O, A = self.obs_dim, self.act_dim
        with stand_ins(specs) as C:
            VN, RMS, Box = (C[(m, n)] for m, n, _ in specs)
            box = lambda n, lo, hi: instance(Box, {"dtype": np.dtype("float32"), "shape": (n,), "low": np.full(n, lo, np.float32),
                                                  "high": np.full(n, hi, np.float32), "np_random": None})
            rms = lambda r: instance(RMS, {"mean": r.mean.cpu().numpy().copy(), "var": r.var.cpu().numpy().copy(),
                                           "count": float(r.count)})
            old_obs = self.old_obs.cpu().numpy() if self.old_obs is not None else np.zeros((self.num_envs, O), np.float32)
            old_rew = self.old_reward.cpu().numpy() if self.old_reward is not None else np.zeros(self.num_envs, np.float32)
            obj = instance(VN, {
                "num_envs": int(self.num_envs), "observation_space": box(O, -self.clip_obs, self.clip_obs) if self.norm_obs
                else box(O, -np.inf, np.inf), "action_space": box(A, -1.0, 1.0), "norm_obs": bool(self.norm_obs),
                "norm_obs_keys": None, "obs_spaces": None, "obs_rms": rms(self.obs_rms), "ret_rms": rms(self.ret_rms),
                "clip_obs": float(self.clip_obs), "clip_reward": float(self.clip_reward), "gamma": float(self.gamma),
                "epsilon": float(self.epsilon), "training": bool(self.training), "norm_reward": bool(self.norm_reward),
                "old_obs": old_obs, "old_reward": old_rew})
            with open(path, "wb") as fh:
                pickle.dump(obj, fh, protocol=4)

    def _load_state(self, st) -> None:
        self.obs_rms.load(**st["obs_rms"])
        self.ret_rms.load(**st["ret_rms"])
        for k in ("clip_obs", "clip_reward", "gamma", "epsilon", "norm_obs", "norm_reward", "training"):
            setattr(self, k, st[k])

    @staticmethod
    def read_pickle(path: str) -> dict:
        """Statistics of an SB3 VecNormalize pickle (read without SB3) or of a file written by save()."""
        with open(path, "rb") as fh:
            obj = _StubUnpickler(fh).load()
        if isinstance(obj, dict):
            return obj
        g = lambda o: {"mean": np.asarray(o.mean), "var": np.asarray(o.var), "count": float(o.count)}
        return {"obs_rms": g(obj.obs_rms), "ret_rms": g(obj.ret_rms), "clip_obs": float(obj.clip_obs),
                "clip_reward": float(obj.clip_reward), "gamma": float(obj.gamma), "epsilon": float(obj.epsilon),
                "norm_obs": bool(obj.norm_obs), "norm_reward": bool(obj.norm_reward), "training": bool(obj.training)}

    @classmethod
    def load(cls, load_path: str, venv) -> "VecNormalize":
        st = cls.read_pickle(load_path)
        self = cls(venv)
        self._load_state(st)
        return self
