"""``MixtureModelBaodingEnv``, batched (/root/reference/src/envs/baoding.py:650-714).

The reference env is the phase-2 Baoding env whose ``reset()`` lets a frozen *base* policy act for the
first ``n_steps_base_model`` (default 20) steps and returns the observation after them, so the learner
only ever sees episodes from step 20 on (the base policy "hands over" a rotating pair of balls).  The
inner steps are plain ``self.step(action)`` calls of the unwrapped env: they advance the goal counter but
are invisible to ``TimeLimit`` / ``Monitor``.

Batched form: after a reset of the whole batch, or after any auto-reset inside ``step_tensor``, the envs
that were just reset run the base phase together through ``myo_batch_step_inner`` (mask = those envs;
the others are untouched).  Cost note: the base phase is ``n_steps_base_model`` extra launches whenever
at least one env finished in a step — with thousands of envs that is every step, i.e. this env is ~20x
slower per learner step than ``CustomMyoBaodingBallsP2``; it is here for coverage of the reference's
curriculum, not as a benchmark configuration.
"""
from __future__ import annotations

from typing import Optional

import torch

from .baoding import BaodingVecEnv


class MixtureModelBaodingVecEnv(BaodingVecEnv):
    def __init__(self, env_name, num_envs, config, *, base_model_path: str, base_env_path: str, base_env_name: str = None,
                 base_env_config: Optional[dict] = None, n_steps_base_model: Optional[int] = None, base_policy=None,
                 base_normalizer=None, **batch_kw):
        super().__init__(env_name, num_envs, config, **batch_kw)
        self.n_steps_base_model = 20 if n_steps_base_model is None else int(n_steps_base_model)
        # load_model_and_env (baoding.py:685-698): RecurrentPPO.load(model_path) + VecNormalize.load(env_path)
        from ..rl.sb3_zip import load_policy
        from ..rl.vec_normalize import VecNormalize
        self.model_base = base_policy if base_policy is not None else load_policy(base_model_path)[0]
        self.model_base.to(self.device).eval()
        if base_normalizer is not None:
            self.env_base = base_normalizer
        else:
            self.env_base = VecNormalize.load(base_env_path, self)
        self.env_base.training = False        # baoding.py:673
        self._inner_done = torch.zeros(num_envs, dtype=torch.uint8, device=self.device)

    @torch.no_grad()
    def _base_phase(self, mask: torch.Tensor) -> None:
        """mask: uint8 [N], envs whose episode has just been reset.  Runs the base policy on them for
        n_steps_base_model inner steps (baoding.py:700-711); their rows of the observation buffer end up
        holding the hand-over observation."""
        N = self.num_envs
        state = self.model_base.initial_state(N, self.device)
        starts = torch.ones(N, device=self.device)
        for _ in range(self.n_steps_base_model):
            act, _, _, state = self.model_base.act(self.env_base.normalize_obs(self._obs), state, starts, deterministic=True)
            act = torch.clamp(act, -1.0, 1.0).to(torch.float32).contiguous()
            self.batch.step_inner(mask, act, self._obs, self._inner_done, self._stream())
            starts = (self._inner_done.bool() & mask.bool()).to(torch.float32)     # episode_starts = dones

    def reset_tensor(self):
        super().reset_tensor()
        self._base_phase(torch.ones(self.num_envs, dtype=torch.uint8, device=self.device))
        return self._obs

    def step_tensor(self, actions):
        out = super().step_tensor(actions)
        done = out[2]
        if bool(done.any()):
            self._base_phase(done.clone())
        return out
