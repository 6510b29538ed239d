"""Batched deterministic evaluation on the device.

Replaces the reference's single-environment evaluation loops — ``src/main_eval.py:86-118`` (reset,
``model.predict(envs.normalize_obs(obs), state, episode_start, deterministic=True)``, step until done,
accumulate reward / length) and the in-training evaluators ``EvaluateLSTM``
(``src/metrics/custom_callbacks.py:19-47``) / SB3 ``EvalCallback`` (``src/main_baoding.py:81-91``),
which cost the reference more than half of its wall time.  Here all episodes run in
parallel on one batched env; every env plays a fixed quota of episodes (as SB3's ``evaluate_policy``
does, so short episodes are not over-represented); rewards are the raw env rewards (the reference
evaluates on an un-normalised env and only normalises the observations).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .. import native


@torch.no_grad()
def evaluate_policy(policy, env, normalizer=None, n_eval_episodes: int = 100, deterministic: bool = True,
                    max_steps: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Play ``n_eval_episodes`` episodes of ``policy`` on the batched ``env`` (a ``BaodingVecEnv``; NOT a
    VecNormalize wrapper — pass the wrapper, or anything with ``normalize_obs``, as ``normalizer``).

    Returns ``{"returns", "lengths", "solved_frac", "truncated"}`` (one entry per episode)."""
    N = env.num_envs
    dev = env.device
    quota = torch.tensor([(n_eval_episodes + i) // N for i in range(N)], device=dev)
    played = torch.zeros(N, dtype=torch.long, device=dev)
    norm = (lambda o: normalizer.normalize_obs(o)) if normalizer is not None else (lambda o: o)
    was_training = getattr(policy, "training", False)
    policy.eval()
    obs = env.reset_tensor()
    starts = torch.ones(N, device=dev)
    state = policy.initial_state(N, dev)
    solved_sum = torch.zeros(N, device=dev)
    j_solved = native.RWD_KEYS.index("solved")
    rets, lens, solved, truncs = [], [], [], []
    horizon = max_steps if max_steps is not None else int(quota.max()) * (getattr(env, "max_episode_steps", 200) + 1) + 1
    for _ in range(horizon):
        actions, _, _, state = policy.act(norm(obs), state, starts, deterministic=deterministic)
        obs, rew, done, trunc, term, comps, ep = env.step_tensor(torch.clamp(actions, -1.0, 1.0))
        solved_sum += comps[:, j_solved]
        dn = done.bool()
        if bool(dn.any()):
            count = dn & (played < quota)            # episodes beyond an env's quota are discarded
            idx = count.nonzero().flatten()
            if idx.numel():
                r, l = ep[idx, 0], ep[idx, 1]
                rets.append(r.cpu().numpy()); lens.append(l.cpu().numpy())
                solved.append((solved_sum[idx] / torch.clamp(l, min=1)).cpu().numpy())
                truncs.append(trunc[idx].bool().cpu().numpy())
                played[idx] += 1
            solved_sum[dn] = 0
        starts = dn.to(torch.float32)
        if bool((played >= quota).all()):
            break
    if was_training:
        policy.train()
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return {"returns": cat(rets, np.float64), "lengths": cat(lens, np.int64), "solved_frac": cat(solved, np.float64),
            "truncated": cat(truncs, bool)}


def summarize(res: Dict[str, np.ndarray]) -> Dict[str, float]:
    """Mean and standard error as printed by the reference every 10 episodes (src/main_eval.py:110-116)."""
    n = max(1, len(res["returns"]))
    return {"episodes": len(res["returns"]), "mean_len": float(np.mean(res["lengths"])) if n else 0.0,
            "len_err": float(np.std(res["lengths"]) / np.sqrt(n)), "mean_rew": float(np.mean(res["returns"])),
            "rew_err": float(np.std(res["returns"]) / np.sqrt(n)), "solved_frac": float(np.mean(res["solved_frac"]))}
