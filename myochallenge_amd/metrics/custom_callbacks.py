"""Callbacks with the reference's names (src/metrics/custom_callbacks.py, SB3 EvalCallback), batched.

They are plain callables ``cb(algo)`` for ``PPO.learn(callback=...)``, called at the end of every rollout and BEFORE the update, i.e.
with the policy every ``_on_step`` of that rollout would have seen.  What a frequency counts follows the class it stands for:

* SB3's ``EvalCallback`` / ``CheckpointCallback`` (/root/reference/src/main_baoding.py:81-98) test ``n_calls % freq == 0`` — ``n_calls``
  = steps of the VECTORISED env, one per ``env.step`` whatever its width (``algo.n_calls``); ``eval_freq=10_000`` is every 10,000
  vec-env steps = 160,000 timesteps on the reference's 16 workers, 41 M timesteps on 4096 envs.  A rollout that crosses one or more
  multiples fires once (the policy is the same at every step of a rollout); checkpoints are named by the timestep count of the
  LAST multiple crossed, as SB3 would have named the last of them.
* the reference's own ``EvaluateLSTM`` (src/metrics/custom_callbacks.py:19-47) tests ``num_timesteps % eval_freq == 0`` — env
  TIMESTEPS, which advance by the number of envs per vec-env step: it fires when a multiple of ``eval_freq`` is among the values
  ``num_timesteps`` took during the rollout.
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import numpy as np

from .evaluation import evaluate_policy


def _rollout_start(algo):
    """(vec-env steps, timesteps) the run had taken BEFORE the rollout that precedes this callback call: what a callback object
    created for a RESUMED run starts counting from.  SB3's BaseCallback.n_calls restarts at 0 with a new callback object; a counter
    that started at 0 against an `algo.n_calls` derived from a checkpoint's num_timesteps fired every callback on the first
    rollout after a resume and named the checkpoint after an old multiple."""
    steps = int(getattr(getattr(algo, "cfg", None), "n_steps", 0))
    return max(0, algo.n_calls - steps), max(0, algo.num_timesteps - steps * _envs_total(algo))


def _calls_crossed(algo, last_calls: int, freq: int, base: int = 0):
    """multiples of `freq` among the vec-env step counts (last_calls, algo.n_calls - base], counted from `base` as SB3 counts from
    the callback's creation; returns (any, last multiple)"""
    freq = max(1, int(freq))
    now = algo.n_calls - base
    last_multiple = (now // freq) * freq
    return last_multiple > last_calls, last_multiple


def _envs_total(algo) -> int:
    return max(1, algo.env.num_envs * getattr(algo, "world", 1))


class EvaluateLSTM:
    """src/metrics/custom_callbacks.py:7-48: every ``eval_freq`` timesteps play ``num_episodes``
    deterministic episodes with the *training* model and the *training* env's normaliser; record the
    mean return under ``name``.  The reference plays them one after another on one env."""

    def __init__(self, eval_freq, eval_env, name, num_episodes=20, log: Optional[Callable[[dict], None]] = None):
        self.eval_freq, self.eval_env, self.name, self.num_episodes, self.log = eval_freq, eval_env, name, num_episodes, log
        self._last_ts = None                       # (set on the first call: the run's timesteps before that rollout — 0 unless resumed)
        self.history = []

    def __call__(self, algo) -> bool:
        # num_timesteps took the values last + N, last + 2 N, ... now (N envs per vec-env step): is a multiple of eval_freq among them?
        if self._last_ts is None:
            self._last_ts = _rollout_start(algo)[1]
        n, now, last = _envs_total(algo), algo.num_timesteps, self._last_ts
        self._last_ts = now
        import math
        lcm = self.eval_freq * n // math.gcd(int(self.eval_freq), n)
        if (now // lcm) * lcm <= last:
            return True
        normalizer = algo.env if hasattr(algo.env, "normalize_obs") else None
        res = evaluate_policy(algo.policy, self.eval_env, normalizer, self.num_episodes, deterministic=True)
        mean = float(np.mean(res["returns"]))
        self.history.append((algo.num_timesteps, mean))
        if self.log is not None:
            self.log({self.name: mean})
        return True


class EnvDumpCallback:
    """src/metrics/custom_callbacks.py:51-62: save ``best_model`` and the training env's normaliser
    (``training_env.save``) into ``save_path`` (used as ``callback_on_new_best``)."""

    def __init__(self, save_path, verbose=0):
        self.save_path, self.verbose = save_path, verbose

    def __call__(self, algo) -> bool:
        os.makedirs(self.save_path, exist_ok=True)
        algo.save(os.path.join(self.save_path, "best_model.zip"))
        if hasattr(algo.env, "save"):
            algo.env.save(os.path.join(self.save_path, "training_env.pkl"))
        return True


class EvalCallback:
    """SB3 ``EvalCallback`` as the reference configures it (src/main_baoding.py:81-91): every ``eval_freq``
    timesteps evaluate ``n_eval_episodes`` deterministic episodes on ``eval_env`` (its own normaliser
    statistics synchronised from the training env first, as SB3's ``sync_envs_normalization``), append to
    ``<log_path>/evaluations.npz`` (keys ``timesteps``, ``results``, ``ep_lengths`` — SB3's format), keep
    the best mean reward and fire ``callback_on_new_best``."""

    def __init__(self, eval_env, callback_on_new_best=None, n_eval_episodes=10, best_model_save_path=None, log_path=None,
                 eval_freq=10_000, deterministic=True, render=False, verbose=1):
        self.eval_env, self.on_best, self.n, self.best_path = eval_env, callback_on_new_best, n_eval_episodes, best_model_save_path
        self.log_path, self.eval_freq, self.deterministic, self.verbose = log_path, eval_freq, deterministic, verbose
        self._last_calls, self._base = 0, None     # (vec-env steps counted from the callback's first rollout, as SB3's n_calls)
        self.best_mean_reward = -np.inf
        self.evaluations_timesteps, self.evaluations_results, self.evaluations_length = [], [], []

    def __call__(self, algo) -> bool:
        if self._base is None:
            self._base = _rollout_start(algo)[0]
        fire, _ = _calls_crossed(algo, self._last_calls, self.eval_freq, self._base)
        self._last_calls = algo.n_calls - self._base
        if not fire:
            return True
        env = self.eval_env
        normalizer = env if hasattr(env, "normalize_obs") else (algo.env if hasattr(algo.env, "normalize_obs") else None)
        if hasattr(env, "obs_rms") and hasattr(algo.env, "obs_rms") and env is not algo.env:     # sync_envs_normalization
            env.obs_rms.load(**algo.env.obs_rms.state())
            env.ret_rms.load(**algo.env.ret_rms.state())
        raw = env.venv if hasattr(env, "venv") else env
        res = evaluate_policy(algo.policy, raw, normalizer, self.n, deterministic=self.deterministic)
        self.evaluations_timesteps.append(algo.num_timesteps)
        self.evaluations_results.append(res["returns"][:self.n])
        self.evaluations_length.append(res["lengths"][:self.n])
        if self.log_path is not None:
            os.makedirs(self.log_path, exist_ok=True)
            np.savez(os.path.join(self.log_path, "evaluations"), timesteps=self.evaluations_timesteps,
                     results=self.evaluations_results, ep_lengths=self.evaluations_length)
        mean = float(np.mean(res["returns"]))
        if self.verbose:
            print(f"Eval num_timesteps={algo.num_timesteps}, episode_reward={mean:.2f} +/- {np.std(res['returns']):.2f}")
        if mean > self.best_mean_reward:
            self.best_mean_reward = mean
            if self.best_path is not None:
                os.makedirs(self.best_path, exist_ok=True)
                algo.save(os.path.join(self.best_path, "best_model.zip"))
            if self.on_best is not None:
                self.on_best(algo)
        return True


class CheckpointCallback:
    """SB3 ``CheckpointCallback`` as the reference configures it (src/main_baoding.py:93-98): every ``save_freq``
    timesteps write ``<save_path>/<name_prefix>_<steps>_steps.zip`` and, with ``save_vecnormalize``,
    ``<name_prefix>_vecnormalize_<steps>_steps.pkl`` (the file names found under
    trained_models/winning_ensemble/base)."""

    def __init__(self, save_freq, save_path, name_prefix="rl_model", save_vecnormalize=False, verbose=0):
        self.save_freq, self.save_path, self.name_prefix = save_freq, save_path, name_prefix
        self.save_vecnormalize, self.verbose = bool(save_vecnormalize) and save_vecnormalize != "False", verbose
        self._last_calls, self._base = 0, None

    def __call__(self, algo) -> bool:
        if self._base is None:
            self._base = _rollout_start(algo)[0]
        fire, last_multiple = _calls_crossed(algo, self._last_calls, self.save_freq, self._base)
        self._last_calls = algo.n_calls - self._base
        if not fire:
            return True
        steps = (self._base + last_multiple) * _envs_total(algo)      # num_timesteps at the (last) vec-env step SB3 would have saved on
        os.makedirs(self.save_path, exist_ok=True)
        path = os.path.join(self.save_path, f"{self.name_prefix}_{steps}_steps.zip")
        algo.save(path)
        if self.save_vecnormalize and hasattr(algo.env, "save"):
            algo.env.save(os.path.join(self.save_path, f"{self.name_prefix}_vecnormalize_{steps}_steps.pkl"))
        if self.verbose:
            print(f"Saving model checkpoint to {path}")
        return True


class TensorboardCallback:
    """src/metrics/custom_callbacks.py:64-81: record the mean of ``info[key]`` (the env's reward-dictionary
    entries, ``info.update(rwd_dict)`` in reorient.py:211) under ``rollout/<key>``.  The batched envs expose the
    last step's reward dictionary as ``env.rwd_dict``; its batch mean is recorded once per rollout."""

    def __init__(self, info_keywords, log=None, verbose=0):
        self.info_keywords, self.log = tuple(info_keywords), log
        self.history = []

    def __call__(self, algo) -> bool:
        env = algo.env.venv if hasattr(algo.env, "venv") else algo.env
        if getattr(env, "track_rwd_dict", None) is False:      # die-reorient env: ask it to keep the shaping terms from now on
            env.track_rwd_dict = True
        rd = getattr(env, "rwd_dict", None)
        if not rd:
            return True
        rec = {"rollout/" + k: float(rd[k].mean()) for k in self.info_keywords if k in rd}
        rec["timesteps"] = algo.num_timesteps
        self.history.append(rec)
        if self.log is not None:
            self.log(rec)
        return True
