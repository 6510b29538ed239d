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

The base phase loops on the GOAL COUNTER, as the reference does (``while self.counter < self.n_steps_base_model``,
baoding.py:704): an env whose reset already took a step (reset-state initialisation calls ``self.step`` once, baoding.py:610-638,
so its counter is 1) takes 19 base-policy steps, not 20 — rows of the block drop out of the inner steps individually.

That form is exact per env — it is the DEFAULT (``pool_size`` = 0), and what evaluation should use: an env's trajectory is a
function of (seed, env index) alone — but LATENCY-bound: 20 sequential env steps are ~15 ms however few envs take them, and
at thousands of envs some episode ends in almost every learner step.  ``pool_size`` > 0 (opt-in, for training throughput;
``"auto"``: 2048 from 4096 envs up) takes the base phase off the learner's critical path: a second batch of ``pool_size`` envs is reset
and played through the base phase IN BULK (20 full-width policy calls + inner steps per refill), and an env of
the learner's batch that finishes an episode receives the whole record of a pool env — its reset state, the
episode's draws and the 20 base-policy steps — through ``myo_batch_copy_envs``.  A learner step then costs the
plain phase-2 step plus (resets per step / pool_size) of a refill.  The hand-over states are the same
distribution as the reference's (reset draws + 20 deterministic base-policy steps), drawn from the pool's own
Philox streams instead of the finishing env's — per-env trajectories then depend on the order in which episodes end (documented
trade: tests/test_gpu_parity.py checks that a pool env's record after the bulk base phase is bit-identical to what the exact
path computes for that env's seed).
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
        if pool_size == "auto":
            pool_size = min(2048, max(256, num_envs // 2)) if num_envs >= 256 else 0
        self.pool_size = 0 if pool_size is None else int(pool_size)
        self._pool = None
        if self.pool_size > 0:
            kw = dict(batch_kw)
            kw["seed"] = int(kw.get("seed", 0)) + 7919
            kw["model"] = self.compiled                         # the same compiled model
            pool_cls = BaodingVecEnv if type(self)._select_device is BaodingVecEnv._select_device else \
                type("_PoolVecEnv", (BaodingVecEnv,), {"_select_device": type(self)._select_device})      # (test builds: the env's own device rule)
            self._pool = pool_cls(env_name, self.pool_size, config, **kw)
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
        self._c_cnt = torch.zeros(C, dtype=torch.int32, device=dev)         # goal counter of the block's envs (the loop condition of baoding.py:704)
        self._task_i = torch.zeros((num_envs, 2), dtype=torch.int32, device=dev)
        # the clipped actions the base policy produced in the last phase, [step, slot, nu]: what an oracle twin is stepped with (tests)
        self._c_act_log = torch.zeros((self.n_steps_base_model, C, self.act_dim), dtype=torch.float32, device=dev)
        st = self.model_base.initial_state(C, dev)
        self._c_state = None if st is None else tuple(t.clone() for t in st)
        self._graph = None
        self._graph_failed = dev.type != "cuda"
        self._closed_pool = False
        self.base_phase_launches = 0          # (replays or eager phases: test / bench bookkeeping)

    @torch.no_grad()
    def _phase_body(self) -> None:
        """n_steps_base_model inner steps of the envs in the compact block (baoding.py:700-711); static tensors only"""
        for k in range(self.n_steps_base_model):
            act, _, _, st = self.model_base.act(self.env_base.normalize_obs(self._c_obs), self._c_state, self._c_starts, deterministic=True)
            act = torch.clamp(act, -1.0, 1.0).to(torch.float32).contiguous()
            self._c_act_log[k].copy_(act)
            # `while self.counter < self.n_steps_base_model` per env: rows whose counter has arrived sit the step out (index -1 = empty
            # slot: their observation, state and episode-start flag stay as they are)
            active = self._c_cnt < self.n_steps_base_model
            idx = torch.where(active, self._c_idx, torch.full_like(self._c_idx, -1))
            self.batch.step_inner_idx(idx, act, self._c_obs, self._c_done, self._stream())
            if st is not None:
                for dst, src in zip(self._c_state, st):
                    m = active.view((1, -1) + (1,) * (dst.dim() - 2)) if dst.dim() >= 2 else active
                    dst.copy_(torch.where(m, src, dst))
            self._c_starts.copy_(torch.where(active, self._c_done.to(torch.float32), self._c_starts))      # episode_starts = dones (baoding.py:707-711)
            self._c_cnt.add_(active.to(torch.int32))

    @torch.no_grad()
    def _refill_pool(self) -> None:
        """fresh episodes for every pool env, played through the base phase at full width (baoding.py:700-711)"""
        pool = self._pool
        obs = pool.reset_tensor()
        S = self.pool_size
        state = self.model_base.initial_state(S, self.device)
        starts = torch.ones(S, device=self.device)
        ti = torch.zeros((S, 2), dtype=torch.int32, device=self.device)
        pool.batch.get_task(ti, None, None, pool._stream())
        cnt = ti[:, 1].clone()                                            # goal counters after the reset (1 where the reset took a step)
        for _ in range(self.n_steps_base_model):
            act, _, _, new_state = self.model_base.act(self.env_base.normalize_obs(obs), state, starts, deterministic=True)
            act = torch.clamp(act, -1.0, 1.0).to(torch.float32).contiguous()
            active = cnt < self.n_steps_base_model                       # `while self.counter < n_steps_base_model`, per env
            pool.batch.step_inner(active.to(torch.uint8), act, obs, self._pool_done, pool._stream())
            if new_state is not None:
                state = tuple(torch.where(active.view((1, -1) + (1,) * (n.dim() - 2)) if n.dim() >= 2 else active, n, o) for n, o in zip(new_state, state))
            starts = torch.where(active, self._pool_done.to(torch.float32), starts)
            cnt = cnt + active.to(torch.int32)
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
            self.batch.get_task(self._task_i, None, None, self._stream())
            self._c_cnt.fill_(self.n_steps_base_model)                   # (empty slots: never active)
            self._c_cnt[:k] = self._task_i[idx.long(), 1]
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
