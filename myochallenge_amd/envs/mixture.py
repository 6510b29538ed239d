"""``MixtureModelBaodingEnv``, batched (/root/reference/src/envs/baoding.py:650-714).

The reference env is the phase-2 Baoding env whose ``reset()`` lets a frozen *base* policy act for the
first ``n_steps_base_model`` (default 20) steps and returns the observation after them, so the learner
only ever sees episodes from step 20 on (the base policy "hands over" a rotating pair of balls).  The
inner steps are plain ``self.step(action)`` calls of the unwrapped env: they advance the goal counter but
are invisible to ``TimeLimit`` / ``Monitor``.

Batched form: after a reset of the whole batch, or after any auto-reset inside ``step_tensor``, the envs
that were just reset run the base phase together.  They are FEW (an episode lasts ~100-200 steps, so ~1 % of
the batch per step), so the phase runs in compact form: their indices go into a fixed block of
``compact_slots`` slots (observations gathered, LSTM state zeroed), the base policy acts on the block and
``myo_batch_step_inner_idx`` steps exactly those envs — ``n_steps_base_model`` x (small policy call + one
launch of ``compact_slots`` workgroups), captured as ONE hipGraph on the GPU, instead of 20 policy calls and
20 launches over the whole batch (round 3: ~20x the cost of a plain phase-2 step; measured now in DESIGN.md §7).
More resets than slots (the first reset of the whole batch) go through the block in chunks.

That form is exact per env but LATENCY-bound: 20 sequential env steps are ~15 ms however few envs take them, and
at thousands of envs some episode ends in almost every learner step.  ``pool_size`` > 0 (the default from 256
envs up) takes the base phase off the learner's critical path: a second batch of ``pool_size`` envs is reset
and played through the base phase IN BULK (20 full-width policy calls + inner steps per refill), and an env of
the learner's batch that finishes an episode receives the whole record of a pool env — its reset state, the
episode's draws and the 20 base-policy steps — through ``myo_batch_copy_envs``.  A learner step then costs the
plain phase-2 step plus (resets per step / pool_size) of a refill.  The hand-over states are the same
distribution as the reference's (reset draws + 20 deterministic base-policy steps), drawn from the pool's own
Philox streams instead of the finishing env's.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .baoding import BaodingVecEnv


class MixtureModelBaodingVecEnv(BaodingVecEnv):
    def __init__(self, env_name, num_envs, config, *, base_model_path: str, base_env_path: str, base_env_name: str = None,
                 base_env_config: Optional[dict] = None, n_steps_base_model: Optional[int] = None, base_policy=None,
                 base_normalizer=None, pool_size: Optional[int] = None, **batch_kw):
        super().__init__(env_name, num_envs, config, **batch_kw)
        self.pool_size = (min(2048, max(256, num_envs // 2)) if num_envs >= 256 else 0) if pool_size is None else int(pool_size)
        self._pool = None
        if self.pool_size > 0:
            kw = dict(batch_kw)
            kw["seed"] = int(kw.get("seed", 0)) + 7919
            kw["model"] = self.compiled                         # the same compiled model
            self._pool = BaodingVecEnv(env_name, self.pool_size, config, **kw)
            self._pool_next = self.pool_size                    # next unused pool env; == pool_size: the pool needs a refill
            self._pool_done = torch.zeros(self.pool_size, dtype=torch.uint8, device=self.device)
            self._pool_range = torch.arange(self.pool_size, dtype=torch.int32, device=self.device)
            self.pool_refills = 0
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
        self.compact_slots = int(min(256, num_envs))
        C, dev = self.compact_slots, self.device
        self._c_idx = torch.full((C,), -1, dtype=torch.int32, device=dev)
        self._c_obs = torch.zeros((C, self.obs_dim), dtype=torch.float32, device=dev)
        self._c_done = torch.zeros(C, dtype=torch.uint8, device=dev)
        self._c_starts = torch.ones(C, dtype=torch.float32, device=dev)
        st = self.model_base.initial_state(C, dev)
        self._c_state = None if st is None else tuple(t.clone() for t in st)
        self._graph = None
        self._graph_failed = dev.type != "cuda"
        self._closed_pool = False
        self.base_phase_launches = 0          # (replays or eager phases: test / bench bookkeeping)

    @torch.no_grad()
    def _phase_body(self) -> None:
        """n_steps_base_model inner steps of the envs in the compact block (baoding.py:700-711); static tensors only"""
        for _ in range(self.n_steps_base_model):
            act, _, _, st = self.model_base.act(self.env_base.normalize_obs(self._c_obs), self._c_state, self._c_starts, deterministic=True)
            act = torch.clamp(act, -1.0, 1.0).to(torch.float32).contiguous()
            self.batch.step_inner_idx(self._c_idx, act, self._c_obs, self._c_done, self._stream())
            if st is not None:
                for dst, src in zip(self._c_state, st):
                    dst.copy_(src)
            self._c_starts.copy_(self._c_done)                        # episode_starts = dones (baoding.py:707-711)

    @torch.no_grad()
    def _refill_pool(self) -> None:
        """fresh episodes for every pool env, played through the base phase at full width (baoding.py:700-711)"""
        pool = self._pool
        obs = pool.reset_tensor()
        S = self.pool_size
        state = self.model_base.initial_state(S, self.device)
        starts = torch.ones(S, device=self.device)
        for _ in range(self.n_steps_base_model):
            act, _, _, state = self.model_base.act(self.env_base.normalize_obs(obs), state, starts, deterministic=True)
            act = torch.clamp(act, -1.0, 1.0).to(torch.float32).contiguous()
            pool.batch.step_inner(None, act, obs, self._pool_done, pool._stream())
            starts = self._pool_done.to(torch.float32)
        self._pool_next = 0
        self.pool_refills += 1

    @torch.no_grad()
    def _hand_over_from_pool(self, idx_all: torch.Tensor) -> None:
        """envs idx_all (int32, just reset) continue as pool envs that have been through reset + base phase"""
        k, lo = int(idx_all.numel()), 0
        while lo < k:
            if self._pool_next >= self.pool_size:
                self._refill_pool()
            take = min(k - lo, self.pool_size - self._pool_next)
            dst, src = idx_all[lo:lo + take].contiguous(), self._pool_range[self._pool_next:self._pool_next + take]
            self.batch.copy_envs_from(self._pool.batch, dst, src, self._stream())
            self._obs[dst.long()] = self._pool._obs[src.long()]
            self._pool_next += take
            lo += take
            self.base_phase_launches += 1

    @torch.no_grad()
    def _base_phase(self, mask: torch.Tensor) -> None:
        """mask: uint8 [N], envs whose episode has just been reset.  Runs the base policy on them for
        n_steps_base_model inner steps; their rows of the observation buffer end up holding the hand-over observation."""
        idx_all = torch.nonzero(mask, as_tuple=False).flatten().to(torch.int32)
        if self._pool is not None:
            return self._hand_over_from_pool(idx_all)
        C = self.compact_slots
        for lo in range(0, int(idx_all.numel()), C):
            idx = idx_all[lo:lo + C]
            k = int(idx.numel())
            self._c_idx.fill_(-1)
            self._c_idx[:k] = idx
            self._c_obs[:k] = self._obs[idx.long()]
            self._c_starts.fill_(1.0)
            if self._c_state is not None:
                for t in self._c_state:
                    t.zero_()
            self._run_phase()
            self._obs[idx.long()] = self._c_obs[:k]
            self.base_phase_launches += 1

    def _run_phase(self) -> None:
        """One base phase of the compact block.  The first phase runs eagerly (it is also the warm-up hipGraph capture needs), the
        second is captured — capture records, it does not execute — and every phase from then on is one graph replay."""
        want_graph = not self._graph_failed and os.environ.get("MYO_MIXTURE_GRAPH", "1") != "0"
        if want_graph and self._graph is None and self.base_phase_launches >= 1:
            try:
                torch.cuda.synchronize(self.device)
                self.batch.bind_constants(self._stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._phase_body()
                self._graph = g
            except Exception as exc:                                  # noqa: BLE001 - any capture problem: stay on the eager path
                import warnings
                warnings.warn(f"MixtureModelBaodingVecEnv: base phase runs eagerly (graph capture failed: {exc!r})")
                self._graph, self._graph_failed = None, True
                torch.cuda.synchronize(self.device)
        if self._graph is not None:
            self.batch.bind_constants(self._stream())
            self._graph.replay()
        else:
            self._phase_body()

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

    def close(self):
        if getattr(self, "_pool", None) is not None and not getattr(self, "_closed_pool", True):
            self._closed_pool = True
            self._pool.close()
        super().close()
