"""Mixture of ensembles with a task classifier, batched — /root/reference/src/eval_mixture_of_ensembles.py.

Reference behaviour (``SuperModel.process_before_action`` :190-211 and ``eval_perf`` :235-330), per episode:
the *base* ensemble (6 LSTM policies, each with its own VecNormalize statistics) acts with the mean of the
members' deterministic actions; the raw observation slice [29:47] of the first 13 steps is collected, at
step index 12 it is standardised and classified; if the classifier says HOLD the *hold* ensemble (5
policies) takes over for the rest of the episode with fresh LSTM states.  Reported: episode length,
return, effort (mean ||act||/na before each step) and the classifier's error against ``env.which_task``.

Here every environment of a batch carries that state machine as tensors, and both ensembles are
evaluated for the whole batch each step (a member is one batched forward).
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch

from .models.classifier import N_OBS_PER_TRIAL, OBS_SLICE, TaskClassifier, load_scaler


class SuperModel:
    def __init__(self, models_base: Sequence, envs_base: Sequence, models_hold: Sequence, envs_hold: Sequence,
                 classifier: TaskClassifier, scaler, num_envs: int, device):
        """models_*: policies (ActorCriticPolicy); envs_*: one normaliser per member (anything with
        ``normalize_obs``, e.g. VecNormalize.load(...)); scaler: (mean, scale) from load_scaler."""
        assert len(models_base) == len(envs_base) and len(models_hold) == len(envs_hold)
        self.models_base, self.envs_base = list(models_base), list(envs_base)
        self.models_hold, self.envs_hold = list(models_hold), list(envs_hold)
        for env in self.envs_base + self.envs_hold:        # :133-135
            env.training = False
            env.norm_reward = False
        self.device = torch.device(device)
        for m in self.models_base + self.models_hold:
            m.to(self.device).eval()
        self.classifier = classifier.to(self.device).eval()
        self.scaler_mean = torch.as_tensor(scaler[0], dtype=torch.float64, device=self.device)
        self.scaler_scale = torch.as_tensor(scaler[1], dtype=torch.float64, device=self.device)
        N, W = num_envs, OBS_SLICE[1] - OBS_SLICE[0]
        self.N = N
        self.obs_for_classifier = torch.zeros((N, N_OBS_PER_TRIAL, W), dtype=torch.float64, device=self.device)
        self.timestep = torch.zeros(N, dtype=torch.long, device=self.device)
        self.use_hold_net = torch.zeros(N, dtype=torch.bool, device=self.device)
        self.just_switched = torch.zeros(N, dtype=torch.bool, device=self.device)
        self.current_task = torch.ones(N, dtype=torch.long, device=self.device)
        self.states_base = [m.initial_state(N, self.device) for m in self.models_base]
        self.states_hold = [m.initial_state(N, self.device) for m in self.models_hold]

    @classmethod
    def load(cls, base_zips, base_pkls, hold_zips, hold_pkls, classifier_pt, scaler_pkl, env):
        """load_model_and_env / load_classifier / load_data_scaler (:88-165) for a batched ``env``."""
        from .rl.sb3_zip import load_policy
        from .rl.vec_normalize import VecNormalize
        clf = TaskClassifier(N_OBS_PER_TRIAL)
        clf.load_state_dict(torch.load(classifier_pt, map_location="cpu"))
        return cls([load_policy(p)[0] for p in base_zips], [VecNormalize.load(p, env) for p in base_pkls],
                   [load_policy(p)[0] for p in hold_zips], [VecNormalize.load(p, env) for p in hold_pkls],
                   clf, load_scaler(scaler_pkl), env.num_envs, env.device)

    @torch.no_grad()
    def process_before_action(self, obs: torch.Tensor, episode_start: torch.Tensor) -> None:
        es = episode_start.bool()
        self.use_hold_net &= ~es                      # :191-195
        self.just_switched &= ~es
        self.timestep[es] = 0
        collect = self.timestep < N_OBS_PER_TRIAL     # :197-198 (stale rows of a new episode are overwritten in order)
        if bool(collect.any()):
            idx = collect.nonzero().flatten()
            self.obs_for_classifier[idx, self.timestep[idx]] = obs[idx, OBS_SLICE[0]:OBS_SLICE[1]].to(torch.float64)
        classify = self.timestep == N_OBS_PER_TRIAL - 1   # :200-207
        if bool(classify.any()):
            idx = classify.nonzero().flatten()
            x = self.obs_for_classifier[idx].reshape(idx.numel(), -1)
            x = ((x - self.scaler_mean) / self.scaler_scale).to(torch.float32)      # scaler.transform, torch.FloatTensor
            task = self.classifier.predict_task(x)
            self.current_task[idx] = task
            hold = idx[task == 0]
            self.use_hold_net[hold] = True
            self.just_switched[hold] = True
        self.timestep += 1

    @torch.no_grad()
    def predict(self, obs: torch.Tensor, episode_start: torch.Tensor, deterministic: bool = True) -> torch.Tensor:
        """Mean action of the active ensemble per env (eval_perf :250-285)."""
        self.process_before_action(obs, episode_start)
        es = episode_start.to(torch.float32)
        hold_start = torch.maximum(es, self.just_switched.to(torch.float32))     # fresh LSTM states on the switch step
        self.just_switched.zero_()

        def ensemble(models, envs, states, starts):
            acts = []
            for i, (m, e) in enumerate(zip(models, envs)):
                a, _, _, states[i] = m.act(e.normalize_obs(obs), states[i], starts, deterministic=deterministic)
                acts.append(a)
            return torch.stack(acts, 0).mean(0)
        a_base = ensemble(self.models_base, self.envs_base, self.states_base, es)
        a_hold = ensemble(self.models_hold, self.envs_hold, self.states_hold, hold_start)
        return torch.where(self.use_hold_net.unsqueeze(-1), a_hold, a_base)


@torch.no_grad()
def eval_perf(eval_env, mixture_model: SuperModel, num_episodes: int = 2000, verbose: bool = True) -> Dict[str, np.ndarray]:
    """Batched eval_perf (:235-345): each env plays its quota of episodes; returns per-episode lengths,
    returns, effort and the classifier's (target, prediction) pairs taken after the 13th step."""
    N, dev = eval_env.num_envs, eval_env.device
    quota = torch.tensor([(num_episodes + i) // N for i in range(N)], device=dev)
    played = torch.zeros(N, dtype=torch.long, device=dev)
    obs = eval_env.reset_tensor()
    starts = torch.ones(N, device=dev)
    step = torch.zeros(N, dtype=torch.long, device=dev)
    eff = torch.zeros(N, dtype=torch.float64, device=dev)
    na = eval_env.act_dim
    lens, perfs, effort, targets, preds, who, who13 = [], [], [], [], [], [], []
    ti = torch.zeros((N, 2), dtype=torch.int32, device=dev)
    for _ in range(int(quota.max()) * (eval_env.max_episode_steps + 1) + 1):
        action = mixture_model.predict(obs, starts)
        eff += torch.linalg.norm(obs[:, -na:].to(torch.float64), dim=-1) / na       # obs_dict["act"] before the step
        which = None
        if bool((step == N_OBS_PER_TRIAL - 1).any()):
            eval_env.batch.get_task(ti, None, None)                                  # which_task of the running episode
            which = ti[:, 0].clone()
        obs, rew, done, trunc, term, comps, ep = eval_env.step_tensor(torch.clamp(action, -1.0, 1.0))
        step += 1
        at13 = (step == N_OBS_PER_TRIAL) & (played < quota)
        if which is not None and bool(at13.any()):
            idx = at13.nonzero().flatten()
            targets.append(torch.clamp(which[idx], 0, 1).cpu().numpy())
            preds.append(mixture_model.current_task[idx].cpu().numpy())
            who13.append(idx.cpu().numpy())
        dn = done.bool()
        if bool(dn.any()):
            idx = (dn & (played < quota)).nonzero().flatten()
            if idx.numel():
                lens.append(ep[idx, 1].cpu().numpy()); perfs.append(ep[idx, 0].cpu().numpy())
                effort.append((eff[idx] / torch.clamp(step[idx], min=1)).cpu().numpy())
                who.append(idx.cpu().numpy())
                played[idx] += 1
            step[dn] = 0
            eff[dn] = 0
        starts = dn.to(torch.float32)
        if bool((played >= quota).all()):
            break
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    res = {"lengths": cat(lens, np.int64), "returns": cat(perfs, np.float64), "effort": cat(effort, np.float64),
           "classifier_targets": cat(targets, np.int64), "classifier_preds": cat(preds, np.int64),
           "env_index": cat(who, np.int64), "classifier_env_index": cat(who13, np.int64)}      # which env of the batch each entry came from
    if verbose and len(res["lengths"]):
        n = len(res["lengths"])
        print(f"Average len: {res['lengths'].mean():.2f} +/- {res['lengths'].std() / np.sqrt(n):.2f}")
        print(f"Average rew: {res['returns'].mean():.2f} +/- {res['returns'].std() / np.sqrt(n):.2f}")
        print(f"Average eff: {res['effort'].mean():.5f} +/- {res['effort'].std():.5f}")
        if len(res["classifier_targets"]):
            err = np.abs(res["classifier_targets"] - res["classifier_preds"]).sum() / len(res["classifier_targets"])
            print(f"Classifier inaccuracy = {err * 100:.1f}%")
    return res
