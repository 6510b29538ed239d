"""PPO rollout/update engine on device tensors (SB3 ``OnPolicyAlgorithm`` / sb3-contrib
``RecurrentPPO`` semantics).

What the reference runs (``RecurrentPPO.learn``; /root/reference/src/train/trainer.py:66-71):
  collect_rollouts: policy forward (no_grad) -> clip -> env.step -> timeout bootstrap
                    ``r += gamma * V(terminal_obs)`` -> buffer.add
  compute_returns_and_advantage: GAE(gamma, lambda) backward scan with episode_starts[t+1]
  train: n_epochs x minibatches; per-minibatch advantage normalisation; clipped surrogate +
         vf_coef * MSE + ent_coef * entropy; clip_grad_norm_; Adam(eps=1e-5)
Here the buffer, the normaliser and the policy stay on the GPU; the env step is one HIP kernel
launch (myo_batch_step).  Multi-GPU: one process per GPU, envs sharded by rank, ONE all-reduce
of a flat fp32 gradient bucket per optimizer step over RCCL (torch.distributed backend "nccl").
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

from .policy import ActorCriticPolicy


@dataclass
class PPOConfig:
    n_steps: int = 128
    batch_size: int = 4096
    n_epochs: int = 10
    learning_rate: float = 3e-4
    clip_range: float = 0.2
    ent_coef: float = 0.0
    vf_coef: float = 0.5
    gamma: float = 0.99
    gae_lambda: float = 0.95
    max_grad_norm: float = 0.5
    normalize_advantage: bool = True
    bf16: bool = True               # autocast the policy GEMMs to bf16 (fp32 master weights)
    sync_adv_moments: bool = False  # all-reduce advantage moments (exact single-process semantics)
    use_graphs: bool = True         # capture one optimizer step in a hipGraph (MLP policy, GPU only)
    # N > 1 ranks: capture the RCCL all-reduce of the flat gradient INSIDE the optimizer hipGraph (forward/backward -> all-reduce ->
    # clip + Adam as ONE replay per minibatch instead of graph / eager collective / graph).  Opt-in (also MYO_GRAPH_ALLREDUCE=1):
    # it saves one launch gap (~10 us of a ~150 us optimizer step) and could only be exercised on a node with >= 2 GPUs, which
    # the build machine does not have — a capture failure there must not cost the scaling run.  Falls back to the eager
    # collective when the capture raises.
    graph_allreduce: bool = False


def compute_gae(rewards, values, episode_starts, last_values, last_dones, gamma, lam):
    """rewards/values/episode_starts [T,N]; SB3 buffer semantics: non-terminal mask for step t is
    1 - episode_starts[t+1] (1 - dones for the last step).  On the GPU this is one HIP kernel
    (myo_gae); the torch loop below is the CPU statement of the same scan."""
    T = rewards.shape[0]
    if rewards.is_cuda and rewards.dtype == torch.float32:
        import ctypes as C
        from .. import native
        lib = native.load()
        adv, ret = torch.empty_like(rewards), torch.empty_like(rewards)
        p = lambda t: C.c_void_p(t.contiguous().data_ptr())
        lv, ld = last_values.float().contiguous(), last_dones.float().contiguous()
        lib.check(lib.L.myo_gae(p(rewards), p(values), p(episode_starts), p(lv), p(ld), T, rewards.shape[1],
                                float(gamma), float(lam), p(adv), p(ret),
                                C.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)))
        return adv, ret
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        if t == T - 1:
            nonterm, nextv = 1.0 - last_dones, last_values
        else:
            nonterm, nextv = 1.0 - episode_starts[t + 1], values[t + 1]
        delta = rewards[t] + gamma * nextv * nonterm - values[t]
        last = delta + gamma * lam * nonterm * last
        adv[t] = last
    return adv, adv + values


class PPO:
    def __init__(self, env, policy: ActorCriticPolicy, cfg: PPOConfig = PPOConfig(), seed: int = 0):
        self.env, self.policy, self.cfg = env, policy, cfg
        self.device = env.device
        self.policy.to(self.device)
        on_gpu = self.device.type == "cuda"
        if not policy.recurrent and (cfg.n_steps * env.num_envs) % min(cfg.batch_size, cfg.n_steps * env.num_envs):
            # SB3 warns here too and then trains on the truncated last minibatch of each epoch; every update path of this
            # class works on fixed-size minibatches (one captured graph) and SKIPS that remainder instead
            import warnings
            warnings.warn(f"rollout of {cfg.n_steps * env.num_envs} samples is not a multiple of batch_size={cfg.batch_size}: the last "
                          f"{(cfg.n_steps * env.num_envs) % cfg.batch_size} samples of every epoch's permutation are not trained on "
                          "(stable-baselines3 would train on them as a truncated minibatch)")
        self.optimizer = torch.optim.Adam(self.policy.parameters(), lr=cfg.learning_rate, eps=1e-5,
                                          capturable=on_gpu, foreach=True if on_gpu else None)
        self._graph = None
        self._t_host = 0
        self._fused = None
        self._fused_rec = None
        self._flat_adam = None
        sde = getattr(policy, "use_sde", False)
        if cfg.use_graphs and on_gpu and not policy.recurrent:      # (gSDE: same fused step with its own loss stage, FusedPPOStep._sde_loss)
            # flat parameter/grad vectors (before any graph captures addresses) + the one-kernel optimiser
            from .. import native
            from .fused_mlp import FlatAdam, flatten_parameters
            self._flat_adam = FlatAdam(flatten_parameters(self.policy), native.load(), cfg.learning_rate,
                                       cfg.max_grad_norm)
            from .fused_mlp import FusedPPOStep
            self._fused = FusedPPOStep(self.policy, native.load(), cfg.clip_range, cfg.ent_coef, cfg.vf_coef)
            if sde and self._fused.merged is None:       # gSDE has a loss stage on the stacked-trunk path only: keep autograd
                self._fused = self._flat_adam = None
                self.optimizer = torch.optim.Adam(self.policy.parameters(), lr=cfg.learning_rate, eps=1e-5, capturable=on_gpu,
                                                  foreach=True if on_gpu else None)
            if self._fused is not None and len(self._fused.half) == 1:       # one bf16 shadow of the flat vector: Adam keeps it in step
                self._flat_adam.shadow = self._fused.half[0]
                self._fused.adam_syncs_shadow = True
                self._fused.refresh_shadow()
        elif cfg.use_graphs and on_gpu and policy.recurrent:
            # recurrent policy: autograd does forward / BPTT, but on flat parameter / gradient vectors so that the
            # whole minibatch step (zero, forward, backward, clip + Adam) is ONE hipGraph (_train_recurrent_graphed)
            from .. import native
            from .fused_mlp import FlatAdam, flatten_parameters
            self._flat_adam = FlatAdam(flatten_parameters(self.policy), native.load(), cfg.learning_rate,
                                       cfg.max_grad_norm)
            if cfg.bf16 and os.environ.get("MYO_RECURRENT_AUTOGRAD") != "1":
                # hand-derived minibatch step (rl/fused_lstm.py); None when the layout has no stacked actor/critic views
                from .fused_lstm import FusedRecurrentPPOStep
                self._fused_rec = FusedRecurrentPPOStep.create(self.policy, native.load(), cfg.clip_range, cfg.ent_coef, cfg.vf_coef)
                if self._fused_rec is not None:          # one bf16 shadow of the flat vector, kept in step by the Adam kernel
                    self._flat_adam.shadow = self._fused_rec.half[0]
                    self._fused_rec.adam_syncs_shadow = True
                    self._fused_rec.refresh_shadow()
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if self._fused is not None and self._flat_adam is not None and self.world == 1 and os.environ.get("MYO_ADAM_SEPARATE") != "1":
            self._fused.adam = self._flat_adam   # nothing is exchanged between gradient and clip: the fused step also squares the gradient
        self.rank = dist.get_rank() if self.world > 1 else 0
        N, T, O, A = env.num_envs, cfg.n_steps, env.obs_dim, env.act_dim
        d = self.device
        self.obs_buf = torch.zeros((T, N, O), device=d)
        self.act_buf = torch.zeros((T, N, A), device=d)
        self.rew_buf = torch.zeros((T, N), device=d)
        self.val_buf = torch.zeros((T, N), device=d)
        self.logp_buf = torch.zeros((T, N), device=d)
        self.start_buf = torch.zeros((T, N), device=d)
        self._last_obs = None
        self._last_starts = torch.ones(N, device=d)
        self._state = policy.initial_state(N, d)
        self._rollout_state0 = None
        self.num_timesteps = 0
        self.n_updates = 0
        self.ep_returns: list = []
        self.ep_lengths: list = []
        self.gen = torch.Generator(device=self.device if on_gpu else "cpu")   # minibatch permutations on the device
        self.gen.manual_seed(seed + 1000 * self.rank)
        nparam = sum(p.numel() for p in self.policy.parameters())
        self._flat_grad = self.policy._flat["g"] if self._flat_adam is not None else torch.zeros(nparam, device=d)
        self.timers: Dict[str, float] = {"rollout": 0.0, "gae": 0.0, "update": 0.0}

    def _autocast(self):
        on = self.cfg.bf16 and self.device.type == "cuda"
        return torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=on)

    # ---------------------------------------------------------------- rollout (hipGraph path)
    def _graphed_rollout(self) -> bool:
        return self.cfg.use_graphs and self.device.type == "cuda" and hasattr(self.env, "step_tensor")

    @torch.no_grad()
    def _rollout_policy_part(self):
        with self._autocast():
            if self.policy.recurrent:
                actions, values, logp, new_state = self.policy.act(self._obs_s, self._state_s, self._starts_s)
                for dst, src in zip(self._state_new, new_state):
                    dst.copy_(src)
            else:
                actions, values, logp, _ = self.policy.act(self._obs_s, None, None)
        self._act_s.copy_(actions); self._val_s.copy_(values); self._logp_s.copy_(logp)
        self._clip_s.copy_(torch.clamp(actions, -1.0, 1.0))

    @torch.no_grad()
    def _rollout_post_part(self, raw):
        cfg = self.cfg
        outs = self._vec.process_step(*raw) if self._vec is not None else raw
        nobs, rew, done, trunc, term, comps, ep = outs
        with self._autocast():   # timeout bootstrap r += gamma V(terminal_obs) where truncated (mask, no host sync)
            if self.policy.recurrent:    # sb3-contrib: critic state after this step, episode_start False [3P-RECALL]
                tv = self.policy.predict_values(term, self._state_new, None)
            else:
                tv = self.policy.predict_values(term)
        rew = rew + cfg.gamma * tv * trunc.to(rew.dtype)
        idx = self._t_idx
        self.obs_buf.index_copy_(0, idx, self._obs_s.unsqueeze(0))
        self.act_buf.index_copy_(0, idx, self._act_s.unsqueeze(0))
        self.rew_buf.index_copy_(0, idx, rew.unsqueeze(0))
        self.val_buf.index_copy_(0, idx, self._val_s.unsqueeze(0))
        self.logp_buf.index_copy_(0, idx, self._logp_s.unsqueeze(0))
        self.start_buf.index_copy_(0, idx, self._starts_s.unsqueeze(0))
        self._obs_s.copy_(nobs)
        self._starts_s.copy_(done.to(torch.float32))
        if self.policy.recurrent:
            for dst, src in zip(self._state_s, self._state_new):
                dst.copy_(src)
        self._t_idx.add_(1).remainder_(cfg.n_steps)

    # -- native rollout step: HIP kernels for policy input, sampling, VecNormalize and buffer writes
    def _native_rollout(self) -> bool:
        fused = self._fused if self._fused is not None else self._fused_rec
        return (self._graphed_rollout() and fused is not None and fused.merged is not None
                and hasattr(self.env, "process_step") and hasattr(self.env, "obs_rms"))

    def _init_native_rollout(self):
        """Per step: graph A (policy input cast, [LSTM step,] stacked trunks, heads, myo_rollout_sample) -> eager
        myo_batch_step -> graph B (myo_vecnorm_step, myo_rollout_advance).  The timeout bootstrap
        r += gamma V(terminal_obs) is applied once per rollout in finish_rollout() from term_buf/trunc_buf (recurrent
        policy: with the critic's LSTM state after each step, kept in crit_h_buf / crit_c_buf)."""
        import ctypes as C
        vec, raw, d, cfg = self.env, self.env.venv, self.device, self.cfg
        N, A, O, T = vec.num_envs, vec.act_dim, vec.obs_dim, cfg.n_steps
        recurrent = self.policy.recurrent
        fused = self._fused_rec if recurrent else self._fused
        lib = fused.lib
        first = vec.reset_tensor() if self._last_obs is None else self._last_obs
        self._obs_s = first.clone().contiguous()
        self._starts_s = self._last_starts.clone()
        self._clip_s = torch.zeros((N, A), device=d)
        self._t_idx = torch.zeros(1, dtype=torch.int32, device=d)
        self._draw = torch.zeros(2, dtype=torch.int64, device=d)
        self._x2 = torch.empty((2, N, O), device=d, dtype=torch.bfloat16)
        self.term_buf = torch.zeros((T, N, O), device=d)
        self.trunc_buf = torch.zeros((T, N), device=d)
        self._vn_work = torch.zeros(((N + 127) // 128) * 2 * (O + 1), dtype=torch.float64, device=d)
        seed = int(self.gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        p = lambda t: C.c_void_p(t.data_ptr())

        # policy call of a step: ONE launch on the matrix cores (myo_ppo_mlp_rollout) where the architecture fits, else policy-input
        # cast + hipBLASLt trunk / head GEMMs + bias/ReLU kernels + myo_rollout_sample
        rdesc = fused.rollout_desc(self._obs_s, seed, self._draw, self._t_idx, self.obs_buf, self.act_buf, self.val_buf,
                                   self.logp_buf, self._clip_s) if os.environ.get("MYO_ROLLOUT_GEMM") != "1" else None
        if recurrent:
            # LSTM state after the previous step, (actor, critic) stacked; the masked copies that enter the step; the step's
            # pre-activations / gates are scratch.  h is bf16 (the matrix cores' operand, as the update computes it); the CELL state is
            # carried in float32 (_cs32 / _cm32) — what stock nn.LSTM carries (/root/reference/src/main_reorient.py:53-71), what
            # policy.predict carries in an evaluation, and what a bf16 round trip per step would erode over a 300-step episode.  The
            # bf16 copy _cs is what the step leaves for the bootstrap buffers (and what the cell-kernel fallback computes in).
            H, bf = self.policy.hidden, torch.bfloat16
            self._hs, self._cs = torch.zeros((2, N, H), device=d, dtype=bf), torch.zeros((2, N, H), device=d, dtype=bf)
            self._cs32 = torch.zeros((2, N, H), device=d, dtype=torch.float32)
            self._hs[0].copy_(self._state[0][0]); self._hs[1].copy_(self._state[2][0])
            self._cs32[0].copy_(self._state[1][0]); self._cs32[1].copy_(self._state[3][0])
            self._cs.copy_(self._cs32)
            self._hm, self._cm = torch.empty_like(self._hs), torch.empty_like(self._cs)
            self._cm32 = torch.empty_like(self._cs32)
            self._lat, self._cn = torch.empty_like(self._hs), torch.empty_like(self._hs)
            self._ws = torch.empty((2, N, 4 * H), device=d, dtype=bf)
            self._bsum = torch.zeros((2, 1, 4 * H), device=d, dtype=bf)
            self.crit_h_buf, self.crit_c_buf = torch.zeros((T, N, H), device=d, dtype=bf), torch.zeros((T, N, H), device=d, dtype=bf)
            Lw = fused.lstm

        sde = getattr(self.policy, "use_sde", False)
        if sde:          # gSDE: myo_rollout_sample_sde leaves actions / log pi in step tensors, the rollout buffers take them by index
            self._act_s, self._logp_s = torch.zeros((N, A), device=d), torch.zeros(N, device=d)
            # the exploration matrices live in a buffer of PPO's own: the captured graph reads it by address (see _reset_sde_noise)
            self._sde_W = torch.zeros((N,) + tuple(self.policy.log_std.shape), device=d)
            self._reset_sde_noise()

        def sample(saved, mean_h, value_h, st):
            """Actions, log pi, value of this step into the rollout buffers (row t_idx) and the clipped actions for the env."""
            if not sde:
                lib.check(lib.L.myo_rollout_sample(p(mean_h), p(value_h), p(self.policy.log_std.data), N, A, seed, p(self._draw),
                                                   p(self._t_idx), p(self.act_buf), p(self.val_buf), p(self.logp_buf),
                                                   p(self._clip_s), 0, st))
                return
            mu, lat = mean_h.float(), saved[-1][0].float().contiguous()      # latent_pi: the actor trunk's output (LSTM output without a trunk)
            lib.check(lib.L.myo_rollout_sample_sde(p(mu), p(lat), p(self._sde_W), p(self.policy.log_std.data), N, lat.shape[1], A,
                                                   p(self._act_s), p(self._clip_s), p(self._logp_s), 0, st))
            t64 = self._t_idx.long()
            self.act_buf.index_copy_(0, t64, self._act_s.unsqueeze(0))
            self.logp_buf.index_copy_(0, t64, self._logp_s.unsqueeze(0))
            self.val_buf.index_copy_(0, t64, value_h.float().view(1, N))

        def part_a_recurrent():
            st = C.c_void_p(torch.cuda.current_stream(d).cuda_stream)
            lib.check(lib.L.myo_rollout_policy_input(p(self._obs_s), N, O, p(self.obs_buf), p(self._x2), 2, p(self._t_idx), st))
            keep = torch.rsub(self._starts_s, 1.0).view(1, N, 1)            # state zeroed where an episode starts
            torch.mul(self._hs, keep, out=self._hm)
            torch.mul(self._cs32, keep, out=self._cm32)
            if fused.step_kernels:       # recurrent product + cell in one launch; nothing kept for a backward pass
                # one projection GEMM for both LSTMs with the bias in its epilogue: row n = [actor 4H | critic 4H], read in place
                gx = torch.addmm(self._bsum.view(8 * H), self._x2[0], Lw["wihh"].view(8 * H, O).t())
                # (cell state in and out in float32; the bf16 copy beside it for the bootstrap buffers)
                lib.check(lib.L.myo_lstm_step_fwd(p(gx), 4 * H, 8 * H, p(self._hm), None, p(Lw["whhh"]), None, 2, N, H,
                                                  p(self._lat), N * H, p(self._hs), p(self._cs), None, None, p(self._cm32), p(self._cs32), st))
            else:                        # (hidden sizes without a fused time-step kernel: GEMM + cell kernel, cell state through bf16)
                self._cm.copy_(self._cm32)
                gx = torch.baddbmm(self._bsum, self._x2, Lw["wihh"].transpose(1, 2))
                gh = torch.bmm(self._hm, Lw["whhh"].transpose(1, 2))
                lib.check(lib.L.myo_lstm_cell_fwd(p(gx), p(gh), p(self._cm), None, 2 * N, N, H, 1, p(self._lat), p(self._hs), p(self._cs),
                                                  p(self._cn), p(self._ws), st))
                self._cs32.copy_(self._cs)
            t64 = self._t_idx.long()
            self.crit_h_buf.index_copy_(0, t64, self._hs[1:2])
            self.crit_c_buf.index_copy_(0, t64, self._cs[1:2])
            saved, mean_h, value_h = fused.trunk_heads(self._lat)
            sample(saved, mean_h, value_h, st)

        def part_a():
            if recurrent:
                part_a_recurrent()
                return
            if rdesc is not None:
                fused.rollout_policy(rdesc)
                return
            st = C.c_void_p(torch.cuda.current_stream(d).cuda_stream)
            lib.check(lib.L.myo_rollout_policy_input(p(self._obs_s), N, O, p(self.obs_buf), p(self._x2), 2, p(self._t_idx), st))
            saved, mean_h, value_h = fused.trunk_heads(self._x2)
            sample(saved, mean_h, value_h, st)

        # N > 1 ranks with a rank-synchronised normaliser: graph B1 (this rank's batch moments) -> eager all-reduce of
        # 2 O + 3 doubles -> graph B2 (running statistics from the global moments, normalise, buffer writes)
        self._vn_sync = self.world > 1 and getattr(vec, "sync_ranks", False) and vec.training
        self._vn_batch = torch.zeros(2 * O + 3, dtype=torch.float64, device=d)

        def part_b1(rawout):
            obs, rew = rawout[:2]
            st = C.c_void_p(torch.cuda.current_stream(d).cuda_stream)
            lib.check(lib.L.myo_vecnorm_batch_moments(p(obs), p(rew), N, O, p(vec.returns), float(vec.gamma), int(vec.training),
                                                      p(self._vn_work), p(self._vn_batch), st))

        def part_b2(rawout):
            obs, rew, done, trunc, term = rawout[:5]
            st = C.c_void_p(torch.cuda.current_stream(d).cuda_stream)
            lib.check(lib.L.myo_vecnorm_finish(
                p(obs), p(rew), p(done), p(trunc), p(term), N, O, p(vec.obs_rms.mean), p(vec.obs_rms.var), p(vec.obs_rms.count),
                p(vec.ret_rms.buf), p(vec.returns), float(vec.epsilon), float(vec.clip_obs), float(vec.clip_reward),
                int(vec.training), int(vec.norm_obs), int(vec.norm_reward), p(self._obs_s), p(self._starts_s), p(self._t_idx),
                p(self.rew_buf), p(self.start_buf), p(self.term_buf), p(self.trunc_buf), p(self._vn_batch), st))
            lib.check(lib.L.myo_rollout_advance(p(self._t_idx), T, p(self._draw), st))

        def part_b(rawout):
            if self._vn_sync:
                part_b1(rawout)
                dist.all_reduce(self._vn_batch)
                part_b2(rawout)
                return
            obs, rew, done, trunc, term = rawout[:5]
            st = C.c_void_p(torch.cuda.current_stream(d).cuda_stream)
            lib.check(lib.L.myo_vecnorm_step(
                p(obs), p(rew), p(done), p(trunc), p(term), N, O, p(vec.obs_rms.mean), p(vec.obs_rms.var), p(vec.obs_rms.count),
                p(vec.ret_rms.buf), p(vec.returns), float(vec.gamma), float(vec.epsilon), float(vec.clip_obs),
                float(vec.clip_reward), int(vec.training), int(vec.norm_obs), int(vec.norm_reward), p(self._obs_s),
                p(self._starts_s), p(self._t_idx), p(self.rew_buf), p(self.start_buf), p(self.term_buf), p(self.trunc_buf),
                p(self._vn_work), st))
            lib.check(lib.L.myo_rollout_advance(p(self._t_idx), T, p(self._draw), st))

        self._raw = raw
        fused.refresh_shadow()
        # eager warm-up of the two halves (library handles, workspaces) WITHOUT stepping the env: part B runs on
        # the env's static output buffers as they are, and everything it touches is put back afterwards, so the
        # recorded rollout starts exactly at the reset
        rawout = (raw._obs, raw._rew, raw._done, raw._trunc, raw._term, raw._comps, raw._ep)
        restore = (self._obs_s, self._starts_s, vec.obs_rms.buf, vec.ret_rms.buf, vec.returns) + ((self._hs, self._cs, self._cs32) if recurrent else ())
        keep = [t.clone() for t in restore]
        if recurrent:
            self._refresh_rollout_lstm()
        side = torch.cuda.Stream(device=d)
        side.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(side):
            part_a()
            part_b(rawout)
        torch.cuda.current_stream(d).wait_stream(side)
        torch.cuda.synchronize(d)
        for t, k in zip(restore, keep):
            t.copy_(k)
        vec.old_obs, vec.old_reward = rawout[0], rawout[1]      # the env's static output buffers
        self._t_idx.zero_()
        self._draw.zero_()
        self._gA, self._gB, self._gB2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), None
        with torch.cuda.graph(self._gA, capture_error_mode="thread_local"):
            part_a()
        if self._vn_sync:
            self._gB2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._gB, capture_error_mode="thread_local"):
                part_b1(rawout)
            with torch.cuda.graph(self._gB2, capture_error_mode="thread_local"):
                part_b2(rawout)
        else:
            with torch.cuda.graph(self._gB, capture_error_mode="thread_local"):
                part_b(rawout)
        self._rollout_ready = True
        self._native = True

    def _refresh_rollout_lstm(self):
        """Per rollout: the summed LSTM biases the native recurrent rollout adds to the input projection."""
        Lw = self._fused_rec.lstm
        torch.add(Lw["bihh"], Lw["bhhh"], out=self._bsum.view(2, -1))

    def _native_state(self):
        """(h_pi, c_pi, h_vf, c_vf) float32 [1,N,H] from the native recurrent rollout's stacked state."""
        f = lambda z, k: z[k:k + 1].float()
        return (f(self._hs, 0), f(self._cs32, 0).clone(), f(self._hs, 1), f(self._cs32, 1).clone())

    def _init_rollout_graphs(self):
        """Per step: graph A (policy inference + sampling) -> eager myo_batch_step (so its HIP events can
        still bracket the kernel) -> graph B (normaliser, timeout bootstrap, buffer writes)."""
        env, d = self.env, self.device
        self._vec = env if hasattr(env, "process_step") else None
        self._raw = env.venv if self._vec is not None else env
        N, A = env.num_envs, env.act_dim
        first = env.reset_tensor() if self._last_obs is None else self._last_obs
        self._obs_s = first.clone()
        self._starts_s = self._last_starts.clone()
        self._act_s, self._clip_s = torch.zeros((N, A), device=d), torch.zeros((N, A), device=d)
        self._val_s, self._logp_s = torch.zeros(N, device=d), torch.zeros(N, device=d)
        self._t_idx = torch.zeros(1, dtype=torch.long, device=d)
        if self.policy.recurrent:               # static LSTM state (before / after the current step)
            self._state_s = tuple(x.clone() for x in self._state)
            self._state_new = tuple(torch.zeros_like(x) for x in self._state)
        side = torch.cuda.Stream(device=d)
        side.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(side):           # eager warm-up (also binds the env's constants)
            for _ in range(2):
                self._rollout_policy_part()
                raw = self._raw.step_tensor(self._clip_s)
                self._rollout_post_part(raw)
        torch.cuda.current_stream(d).wait_stream(side)
        torch.cuda.synchronize(d)
        self._t_idx.zero_()
        self._raw_static = raw                  # the env returns views of its own (static) buffers
        self._gA, self._gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._gA, capture_error_mode="thread_local"):
            self._rollout_policy_part()
        with torch.cuda.graph(self._gB, capture_error_mode="thread_local"):          # own pool: graph B keeps tensors alive across replays
            self._rollout_post_part(self._raw_static)
        self._rollout_ready = True

    def rollout_step(self) -> None:
        """One environment step for the whole batch (graphed path only)."""
        if not getattr(self, "_rollout_ready", False):
            if self._native_rollout():
                self._init_native_rollout()
            else:
                self._init_rollout_graphs()
        if hasattr(self.env, "begin_rollout"):
            self.env.begin_rollout()         # (per-rollout normaliser synchronisation: snapshot at the first step; a no-op otherwise)
        if self.policy.recurrent:
            if self._t_host == 0:      # LSTM state at the start of the rollout: initial state of the update's sequences
                self._rollout_state0 = self._native_state() if getattr(self, "_native", False) else tuple(s.clone() for s in self._state_s)
            self._t_host = (self._t_host + 1) % self.cfg.n_steps
        self._gA.replay()
        raw = self._raw.step_tensor(self._clip_s)
        self._log_episodes(raw[2], raw[6])
        self._gB.replay()
        if getattr(self, "_gB2", None) is not None:      # rank-synchronised normaliser (see _init_native_rollout)
            # (host time of the eager collective between the two graph replays — enqueue + whatever the call blocks on — accumulated for
            #  the bench line of an N > 1 run: `normalizer_allreduce_ms_per_step`)
            t0 = time.perf_counter()
            dist.all_reduce(self._vn_batch)
            self.vn_allreduce_seconds = getattr(self, "vn_allreduce_seconds", 0.0) + (time.perf_counter() - t0)
            self.vn_allreduce_calls = getattr(self, "vn_allreduce_calls", 0) + 1
            self._gB2.replay()

    def _log_episodes(self, done, ep) -> None:
        """Monitor statistics (SB3's ep_info_buffer): the (raw return, length) the kernel reports for every env that finished an
        episode in this step go into row t of a [T, N, 2] log, NaN elsewhere — one small launch per step; episode_stats() reads it."""
        T, N = self.cfg.n_steps, self.env.num_envs
        if getattr(self, "_ep_log", None) is None or self._ep_log.shape[:2] != (T, N):
            self._ep_log = torch.full((T, N, 2), float("nan"), device=self.device)
            self._ep_t = 0
            from collections import deque
            self.ep_info_buffer = deque(maxlen=100)
        if ep is None:
            return
        torch.where(done.view(torch.bool).unsqueeze(1), ep, self._ep_nan, out=self._ep_log[self._ep_t % T])
        self._ep_t += 1

    @property
    def _ep_nan(self):
        if getattr(self, "_ep_nan_t", None) is None:
            self._ep_nan_t = torch.full((1, 1), float("nan"), device=self.device)
        return self._ep_nan_t

    def episode_stats(self) -> Dict[str, float]:
        """rollout/ep_rew_mean and rollout/ep_len_mean as SB3 logs them: means over the last 100 finished episodes (Monitor's raw,
        un-normalised returns; /root/reference/src/main_baoding.py runs under SB3's logger).  Reads the rollout's episode log once
        (one host synchronisation per call)."""
        log = getattr(self, "_ep_log", None)
        if log is not None and self._ep_t > 0:
            T = log.shape[0]
            rows = min(self._ep_t, T)
            order = [(self._ep_t - rows + k) % T for k in range(rows)]               # oldest step first
            x = log[order].reshape(-1, 2)
            fin = x[~torch.isnan(x[:, 0])]
            if fin.shape[0] > 100:
                fin = fin[-100:]
            for r, l in fin.cpu().tolist():
                self.ep_info_buffer.append((r, l))
            log.fill_(float("nan"))
            self._ep_t = 0
        buf = getattr(self, "ep_info_buffer", None)
        if not buf:
            return {}
        return {"rollout/ep_rew_mean": float(sum(r for r, _ in buf) / len(buf)), "rollout/ep_len_mean": float(sum(l for _, l in buf) / len(buf))}

    def finish_rollout(self) -> None:
        if hasattr(self.env, "end_rollout"):
            self.env.end_rollout()
        with torch.no_grad(), self._autocast():
            if getattr(self, "_native", False) and not self.policy.recurrent:
                # timeout bootstrap for the whole rollout at once: r += gamma V(terminal_obs) where truncated
                T, N = self.trunc_buf.shape
                tv = self.policy.predict_values(self.term_buf.view(T * N, -1)).view(T, N)
                self.rew_buf.add_(self.cfg.gamma * tv * self.trunc_buf)
            elif getattr(self, "_native", False):
                # same with the critic's LSTM state after each step, episode_start False (sb3-contrib [3P-RECALL]); time limits are
                # rare events (one step in max_episode_steps), so the pass is skipped when the rollout has none (one host sync)
                self._state_s = self._native_state()
                if bool(self.trunc_buf.any()):
                    T, N = self.trunc_buf.shape
                    H = self.policy.hidden
                    st = (None, None, self.crit_h_buf.view(1, T * N, H).float(), self.crit_c_buf.view(1, T * N, H).float())
                    tv = self.policy.predict_values(self.term_buf.view(T * N, -1), st, None).view(T, N)
                    self.rew_buf.add_(self.cfg.gamma * tv * self.trunc_buf)
            if self.policy.recurrent:
                self._last_values = self.policy.predict_values(self._obs_s, self._state_s, self._starts_s)
                self._state = self._state_s
            else:
                self._last_values = self.policy.predict_values(self._obs_s)
        self._last_obs, self._last_starts = self._obs_s, self._starts_s

    def _reset_sde_noise(self) -> None:
        """gSDE: new exploration matrices for the rollout.  On the HIP-kernel rollout they are drawn INTO self._sde_W, the buffer whose
        address the captured graph holds; whatever policy.act did to policy.exploration_mat in between (an evaluation on another batch
        size) cannot move it."""
        pol = self.policy
        gen = self.gen if self.gen.device == pol.log_std.device else None
        W = getattr(self, "_sde_W", None)
        if W is not None and W.shape[1:] == tuple(pol.log_std.shape):
            pol.reset_noise(W.shape[0], gen, out=W)
        else:
            pol.reset_noise(self.env.num_envs, gen)

    # ---------------------------------------------------------------- rollout
    @torch.no_grad()
    def collect_rollouts(self) -> None:
        cfg, env, pol = self.cfg, self.env, self.policy
        if getattr(pol, "use_sde", False):          # SB3: policy.reset_noise(env.num_envs) at the start of a rollout
            self._reset_sde_noise()
        if self._graphed_rollout():
            if self._fused is not None:
                self._fused.refresh_shadow()        # rollout inference runs on the bf16 shadow weights
            if self._fused_rec is not None:
                self._fused_rec.refresh_shadow()
                if getattr(self, "_native", False):
                    self._refresh_rollout_lstm()
            for _ in range(cfg.n_steps):
                self.rollout_step()
            self.finish_rollout()
            self.num_timesteps += cfg.n_steps * env.num_envs * self.world
            return
        if self._last_obs is None:
            self._last_obs = env.reset_tensor().clone()
        if pol.recurrent:
            self._rollout_state0 = tuple(s.clone() for s in self._state)
        if hasattr(env, "begin_rollout"):
            env.begin_rollout()
        for t in range(cfg.n_steps):
            obs, starts = self._last_obs, self._last_starts
            with self._autocast():
                actions, values, logp, new_state = pol.act(obs, self._state, starts)
            clipped = torch.clamp(actions, -1.0, 1.0)
            nobs, rew, done, trunc, term, comps, ep = env.step_tensor(clipped)
            self._log_episodes(done, ep)
            rew = rew.clone()
            if bool(trunc.any()):   # timeout bootstrap: terminal observation, critic state after this step, no episode start
                with self._autocast():
                    tv = pol.predict_values(term, new_state, None)
                rew = rew + cfg.gamma * tv * trunc.to(rew.dtype)
            self.obs_buf[t], self.act_buf[t], self.rew_buf[t] = obs, actions, rew
            self.val_buf[t], self.logp_buf[t], self.start_buf[t] = values, logp, starts
            self._state = new_state
            self._last_obs = nobs.clone()
            self._last_starts = done.to(torch.float32)
        self.num_timesteps += cfg.n_steps * env.num_envs * self.world
        if hasattr(env, "end_rollout"):
            env.end_rollout()
        with self._autocast():
            self._last_values = pol.predict_values(self._last_obs, self._state, self._last_starts)

    # ---------------------------------------------------------------- update
    def _allreduce_grads(self) -> None:
        if self.world == 1:
            return
        off = 0
        for p in self.policy.parameters():
            n = p.numel()
            self._flat_grad[off:off + n] = p.grad.reshape(-1) if p.grad is not None else 0
            off += n
        dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
        self._flat_grad /= self.world
        off = 0
        for p in self.policy.parameters():
            n = p.numel()
            p.grad = self._flat_grad[off:off + n].view_as(p).clone()
            off += n

    def _loss(self, values, logp, entropy, old_logp, adv, returns):
        cfg = self.cfg
        if cfg.normalize_advantage and adv.numel() > 1:
            if cfg.sync_adv_moments and self.world > 1:
                m = torch.stack([adv.sum(), (adv * adv).sum(), torch.tensor(float(adv.numel()), device=adv.device)])
                dist.all_reduce(m)
                mean = m[0] / m[2]
                std = torch.sqrt(torch.clamp(m[1] / m[2] - mean * mean, min=0.0) * m[2] / (m[2] - 1))
            else:
                mean, std = adv.mean(), adv.std()
            adv = (adv - mean) / (std + 1e-8)
        ratio = torch.exp(logp - old_logp)
        pl = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - cfg.clip_range, 1 + cfg.clip_range)).mean()
        vl = torch.nn.functional.mse_loss(values, returns)
        ent = entropy.mean() if torch.is_tensor(entropy) and entropy.dim() > 0 else entropy      # gSDE: per sample
        return pl + cfg.ent_coef * (-ent) + cfg.vf_coef * vl, pl.detach(), vl.detach()

    def train(self) -> Dict[str, float]:
        cfg, pol = self.cfg, self.policy
        T, N = cfg.n_steps, self.env.num_envs
        adv, ret = compute_gae(self.rew_buf, self.val_buf, self.start_buf, self._last_values,
                               self._last_starts, cfg.gamma, cfg.gae_lambda)
        stats = {}
        if self._fused is not None:
            pl, vl = self._train_graphed(adv, ret)
        elif not pol.recurrent:
            B = T * N
            obs, act = self.obs_buf.view(B, -1), self.act_buf.view(B, -1)
            oldlp, advf, retf = self.logp_buf.view(B), adv.view(B), ret.view(B)
            bs = min(cfg.batch_size, B)
            for _ in range(cfg.n_epochs):
                perm = torch.randperm(B, generator=self.gen, device=self.device)
                for s in range(0, B - bs + 1, bs):
                    idx = perm[s:s + bs]
                    with self._autocast():
                        v, lp, ent = pol.evaluate_actions(obs[idx], act[idx])
                    loss, pl, vl = self._loss(v, lp, ent, oldlp[idx], advf[idx], retf[idx])
                    self.optimizer.zero_grad(set_to_none=True)
                    loss.backward()
                    self._allreduce_grads()
                    torch.nn.utils.clip_grad_norm_(pol.parameters(), cfg.max_grad_norm)
                    self.optimizer.step()
                    self.n_updates += 1
        elif self._flat_adam is not None:
            pl, vl = self._train_recurrent_graphed(adv, ret)
        else:
            # sequences = whole rollouts of a subset of envs, initial LSTM state = state at rollout start
            envs_per_mb = max(1, min(N, cfg.batch_size // T))
            for _ in range(cfg.n_epochs):
                perm = torch.randperm(N, generator=self.gen, device=self.device)
                for s in range(0, N - envs_per_mb + 1, envs_per_mb):
                    idx = perm[s:s + envs_per_mb]
                    st0 = tuple(x[:, idx] for x in self._rollout_state0)
                    with self._autocast():
                        v, lp, ent = pol.evaluate_actions(self.obs_buf[:, idx], self.act_buf[:, idx], st0,
                                                          self.start_buf[:, idx])
                    loss, pl, vl = self._loss(v.reshape(-1), lp.reshape(-1), ent, self.logp_buf[:, idx].reshape(-1),
                                              adv[:, idx].reshape(-1), ret[:, idx].reshape(-1))
                    self.optimizer.zero_grad(set_to_none=True)
                    loss.backward()
                    self._allreduce_grads()
                    torch.nn.utils.clip_grad_norm_(pol.parameters(), cfg.max_grad_norm)
                    self.optimizer.step()
                    self.n_updates += 1
        stats.update(policy_loss=float(pl), value_loss=float(vl), n_updates=self.n_updates)
        return stats

    # ---------------------------------------------------------------- hipGraph-captured minibatch step
    def _mb_forward_backward(self):
        g, cfg = self._gs, self.cfg
        idx = g["idx"]
        pl, vl = self._fused.run_indexed(g["obs"], g["act"], g["oldlp"], g["adv"], g["ret"], idx)
        g["pl"], g["vl"] = pl, vl                 # views of the loss kernel's accumulator: no copies      # gradients land in the flat vector (p.grad are views of it)

    def _mb_apply(self):
        self._flat_adam.step(1.0 / self.world)    # all-reduce SUM ran in place on the flat gradient

    def _build_graphs(self, B, bs):
        d = self.device
        self._gs = {"idx": torch.zeros(bs, dtype=torch.long, device=d),
                    "obs": self.obs_buf.view(B, -1), "act": self.act_buf.view(B, -1), "oldlp": self.logp_buf.view(B),
                    "adv": torch.zeros(B, device=d), "ret": torch.zeros(B, device=d),
                    "pl": torch.zeros((), device=d), "vl": torch.zeros((), device=d)}
        self._gs["idx"].copy_(torch.arange(bs, device=d))
        side = torch.cuda.Stream(device=d)
        # warm-up and capture must not move the parameters.  The snapshot is taken BEFORE the side stream is told to wait: its copies run
        # on the current stream, and a warm-up optimizer step on the side stream that overtook them left some ranks restoring a state two
        # Adam steps ahead (round 5: replicas of >= 4 ranks sharing a GPU differed; found by the config-D-shape test)
        snap = self._flat_adam.snapshot()
        side.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(side):       # eager warm-up (library handles, workspaces)
            for _ in range(2):
                self._mb_forward_backward()
                if self.world > 1:
                    dist.all_reduce(self._flat_grad)
                self._mb_apply()
        torch.cuda.current_stream(d).wait_stream(side)
        torch.cuda.synchronize(d)
        self._graph_fb, self._graph_ap = torch.cuda.CUDAGraph(), None
        import os
        # The gradient all-reduce INSIDE the optimizer hipGraph (forward / backward -> RCCL all-reduce -> clip + Adam, one replay per
        # minibatch): opt-in, nccl (= RCCL) only.  There is no in-process fallback from a failed capture: measured on a one-GPU box with a
        # backend that synchronises inside the capture (gloo), the capture is invalidated, capture_end throws inside torch's graph
        # destructor and the process dies (round 5, tests/test_gpu_parity.py).  So a backend that cannot be captured is refused up front
        # (warning, eager collective between two graphs — the tested default), and a capture failure on nccl is raised as an error.
        asked = self.world > 1 and (self.cfg.graph_allreduce or os.environ.get("MYO_GRAPH_ALLREDUCE") == "1")
        want_inside = asked and dist.get_backend() == "nccl"
        if asked and not want_inside:
            import warnings
            warnings.warn(f"graph_allreduce needs the nccl (RCCL) backend, this process group is {dist.get_backend()!r}: "
                          "using the eager all-reduce between the two optimizer graphs")
        self._allreduce_in_graph = False
        if want_inside:
            g_all = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g_all, capture_error_mode="thread_local"):
                    self._mb_forward_backward()
                    dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
                    self._mb_apply()
            except Exception as exc:        # noqa: BLE001
                raise RuntimeError("capturing the RCCL gradient all-reduce inside the optimizer hipGraph failed; run without "
                                   "graph_allreduce / MYO_GRAPH_ALLREDUCE (the eager collective between two graphs)") from exc
            self._graph_fb, self._allreduce_in_graph = g_all, True
        if not self._allreduce_in_graph:
            with torch.cuda.graph(self._graph_fb, capture_error_mode="thread_local"):
                self._mb_forward_backward()
                if self.world == 1:             # single GPU: forward, backward and optimiser in ONE graph
                    self._mb_apply()
            if self.world > 1:                  # the RCCL all-reduce runs between the two graphs
                self._graph_ap = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph_ap, capture_error_mode="thread_local"):
                    self._mb_apply()
        # One GPU: a whole EPOCH in one graph — every minibatch's index copy (from a permutation buffer filled before the replay), forward /
        # backward and optimizer step: one replay instead of B / bs, no launch gap between two minibatch steps and none around the index
        # copy (13 us + a 5 us copy kernel per 135 us step at config B).  Same kernels in the same order as the per-step graph
        # (MYO_EPOCH_GRAPH=0), so the parameters come out bit-identical (tests/test_gpu_parity.py).
        # The capture unrolls its minibatch steps, so it is CHUNKED: an SB3-style batch_size of 64 on 4096 x 64 samples is
        # 4,096 steps an epoch — one graph of that would be hundreds of thousands of nodes to capture, instantiate and hold.  A chunk
        # graph holds at most MYO_EPOCH_GRAPH_STEPS (64) steps, reads its minibatches from the first k * bs entries of a chunk-sized
        # window buffer, and is replayed ceil(n / k) times per epoch (the last, shorter chunk: the per-step graph).
        self._graph_epoch = None
        n_mb = B // bs
        if self.world == 1 and os.environ.get("MYO_EPOCH_GRAPH") != "0" and n_mb > 1:
            k = max(1, min(n_mb, int(os.environ.get("MYO_EPOCH_GRAPH_STEPS", "64"))))
            self._epoch_chunk = k
            self._gs["perm"] = torch.zeros(k * bs, dtype=torch.long, device=d)
            self._gs["perm"].copy_(torch.arange(k * bs, device=d))
            torch.cuda.synchronize(d)
            self._graph_epoch = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_epoch, capture_error_mode="thread_local"):
                for s in range(0, k * bs, bs):
                    self._gs["idx"].copy_(self._gs["perm"][s:s + bs])
                    self._mb_forward_backward()
                    self._mb_apply()
        self._flat_adam.restore(snap)
        self._graph = (B, bs)

    def _train_graphed(self, adv, ret):
        cfg = self.cfg
        B = cfg.n_steps * self.env.num_envs
        bs = min(cfg.batch_size, B)
        # advantage moments of a minibatch: computed by the gather kernel (this rank's rows), or handed to it when they
        # must be the moments of the GLOBAL minibatch (sync_adv_moments, N > 1) or when advantages are not normalised
        ext = (cfg.sync_adv_moments and self.world > 1) or not cfg.normalize_advantage
        if self._fused.external_adv_stats != ext:
            self._fused.external_adv_stats = ext
            self._graph = None
        if self._graph != (B, bs):
            self._build_graphs(B, bs)
        g = self._gs
        g["adv"].copy_(adv.view(B)); g["ret"].copy_(ret.view(B))
        if not cfg.normalize_advantage:
            self._fused.stats.copy_(torch.tensor([0.0, 1.0], device=self.device))
        if getattr(self, "_graph_epoch", None) is not None and not ext:
            k, n_mb = self._epoch_chunk, B // bs
            for _ in range(cfg.n_epochs):
                perm = torch.randperm(B, generator=self.gen, device=self.device)
                done = 0
                while n_mb - done >= k:                        # whole chunks: one replay each
                    g["perm"].copy_(perm[done * bs:(done + k) * bs])
                    self._graph_epoch.replay()
                    done += k
                for j in range(done, n_mb):                    # what is left of the epoch: the per-step graph
                    g["idx"].copy_(perm[j * bs:(j + 1) * bs])
                    self._graph_fb.replay()
                self.n_updates += n_mb
            self._fused.refresh_shadow()
            return g["pl"], g["vl"]
        for _ in range(cfg.n_epochs):
            perm = torch.randperm(B, generator=self.gen, device=self.device)
            for s in range(0, B - bs + 1, bs):
                g["idx"].copy_(perm[s:s + bs])
                if ext and cfg.normalize_advantage:
                    a = g["adv"][perm[s:s + bs]].double()
                    m = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(bs), dtype=torch.float64, device=self.device)])
                    dist.all_reduce(m)
                    mean = m[0] / m[2]
                    std = torch.sqrt(torch.clamp(m[1] / m[2] - mean * mean, min=0.0) * m[2] / (m[2] - 1))
                    self._fused.stats.copy_(torch.stack([mean, std]).float())
                self._graph_fb.replay()
                if self.world > 1 and not self._allreduce_in_graph:
                    dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
                    if self._dp_check:
                        self._dp_compare("gradient after the all-reduce", self._flat_grad)
                        self._dp_compare("parameters before the optimizer step", self._flat_adam.flat["p"])
                        self._dp_compare("Adam m before the optimizer step", self._flat_adam.m)
                        self._dp_compare("Adam v before the optimizer step", self._flat_adam.v)
                        self._dp_compare("Adam step counter before the optimizer step", self._flat_adam._step)
                    self._graph_ap.replay()
                    if self._dp_check:
                        self._dp_compare("gradient after the optimizer step", self._flat_grad)
                        self._dp_compare("|g|^2 partial sums", self._flat_adam.sq[:64].contiguous())
                        self._dp_compare("parameters after the optimizer step", self._flat_adam.flat["p"])
                self.n_updates += 1
        self._fused.refresh_shadow()     # rollout inference reads the bf16 shadow weights
        return g["pl"], g["vl"]

    _dp_check = bool(__import__("os").environ.get("MYO_DP_CHECK"))

    def _dp_compare(self, what: str, x: torch.Tensor) -> None:
        """MYO_DP_CHECK=1 (developer diagnostic of the N > 1 path): a bit-level checksum of `x` must be the same on every rank."""
        cs = x.detach().contiguous().view(torch.int32).to(torch.int64).sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if int(lo[0]) != int(hi[0]) and dist.get_rank() == 0:
            print(f"MYO_DP_CHECK: {what} differs across ranks at optimizer step {self.n_updates}", flush=True)

    # ---------------------------------------------------------------- recurrent policy: one hipGraph per minibatch step
    def _rec_forward_backward(self):
        g = self._rg
        idx = g["idx"]
        if self._fused_rec is not None:
            pl, vl = self._fused_rec.run_sequences(self.obs_buf, self.act_buf, self.start_buf, self.logp_buf, g["adv"], g["ret"],
                                                   g["h0"], g["c0"], idx)
            g["pl"], g["vl"] = pl, vl             # views of the loss kernel's accumulator
            return
        self._flat_grad.zero_()
        st0 = tuple(x.index_select(1, idx) for x in g["state0"])
        sel = lambda buf: buf.index_select(1, idx)
        with self._autocast():
            v, lp, ent = self.policy.evaluate_actions(sel(self.obs_buf), sel(self.act_buf), st0, sel(self.start_buf))
        loss, pl, vl = self._loss(v.reshape(-1), lp.reshape(-1), ent, sel(self.logp_buf).reshape(-1),
                                  sel(g["adv"]).reshape(-1), sel(g["ret"]).reshape(-1))
        loss.backward()                       # p.grad are views of the flat gradient: accumulated in place
        g["pl"].copy_(pl); g["vl"].copy_(vl)

    def _build_recurrent_graphs(self, T, N, m):
        d = self.device
        self._rg = {"idx": torch.arange(m, device=d), "state0": tuple(torch.zeros_like(x) for x in self._rollout_state0),
                    "adv": torch.zeros((T, N), device=d), "ret": torch.zeros((T, N), device=d),
                    "pl": torch.zeros((), device=d), "vl": torch.zeros((), device=d)}
        if self._fused_rec is not None:           # stacked (actor, critic) LSTM state of the rollout start
            H = self.policy.hidden
            self._rg["h0"], self._rg["c0"] = torch.zeros((2, N, H), device=d), torch.zeros((2, N, H), device=d)
        side = torch.cuda.Stream(device=d)
        snap = self._flat_adam.snapshot()        # (before the side stream waits: see _build_graphs)
        side.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._rec_forward_backward()
                if self.world > 1:
                    dist.all_reduce(self._flat_grad)
                self._mb_apply()
        torch.cuda.current_stream(d).wait_stream(side)
        torch.cuda.synchronize(d)
        self._rgraph_fb, self._rgraph_ap = torch.cuda.CUDAGraph(), None
        with torch.cuda.graph(self._rgraph_fb, capture_error_mode="thread_local"):
            self._rec_forward_backward()
            if self.world == 1:
                self._mb_apply()
        if self.world > 1:
            self._rgraph_ap = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._rgraph_ap, capture_error_mode="thread_local"):
                self._mb_apply()
        self._flat_adam.restore(snap)
        self._rgraph = (T, N, m)

    def _rec_stage(self, adv, ret, T, N, m):
        """Graphs for this shape (built on first use) and the update's inputs copied into their static tensors."""
        cfg = self.cfg
        if self._fused_rec is not None:
            ext = (cfg.sync_adv_moments and self.world > 1) or not cfg.normalize_advantage
            if self._fused_rec.external_adv_stats != ext:
                self._fused_rec.external_adv_stats = ext
                self._rgraph = None
        if getattr(self, "_rgraph", None) != (T, N, m):
            self._build_recurrent_graphs(T, N, m)
        g = self._rg
        g["adv"].copy_(adv); g["ret"].copy_(ret)
        for dst, src in zip(g["state0"], self._rollout_state0):
            dst.copy_(src)
        fr = self._fused_rec
        if fr is not None:
            fr.refresh_shadow()                   # (parameters may have been loaded since the last update)
            hp, cp, hv, cv = self._rollout_state0
            g["h0"][0].copy_(hp[0]); g["h0"][1].copy_(hv[0]); g["c0"][0].copy_(cp[0]); g["c0"][1].copy_(cv[0])
            if not cfg.normalize_advantage:
                fr.stats.copy_(torch.tensor([0.0, 1.0], device=self.device))
        return g

    def _train_recurrent_graphed(self, adv, ret):
        """Same minibatches as the eager recurrent path (whole rollouts of a random subset of envs, LSTM state of
        the rollout start), replayed from a graph: BPTT over n_steps is thousands of small launches."""
        cfg = self.cfg
        T, N = cfg.n_steps, self.env.num_envs
        m = max(1, min(N, cfg.batch_size // T))
        g, fr = self._rec_stage(adv, ret, T, N, m), self._fused_rec
        for _ in range(cfg.n_epochs):
            perm = torch.randperm(N, generator=self.gen, device=self.device)
            for s in range(0, N - m + 1, m):
                g["idx"].copy_(perm[s:s + m])
                if fr is not None and fr.external_adv_stats and cfg.normalize_advantage:      # moments of the GLOBAL minibatch
                    a = g["adv"][:, perm[s:s + m]].double()
                    mom = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(a.numel()), dtype=torch.float64, device=self.device)])
                    dist.all_reduce(mom)
                    mean = mom[0] / mom[2]
                    std = torch.sqrt(torch.clamp(mom[1] / mom[2] - mean * mean, min=0.0) * mom[2] / (mom[2] - 1))
                    fr.stats.copy_(torch.stack([mean, std]).float())
                self._rgraph_fb.replay()
                if self.world > 1:
                    dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
                    self._rgraph_ap.replay()
                self.n_updates += 1
        return g["pl"], g["vl"]

    # ---------------------------------------------------------------- persistence (SB3 zip layout)
    def _optimizer_state_dict(self) -> dict:
        """torch.optim.Adam state_dict layout (what SB3 stores in policy.optimizer.pth), parameter order of
        ``policy.parameters()`` as SB3 builds its optimiser."""
        cfg = self.cfg
        params = list(self.policy.parameters())
        state = {}
        if self._flat_adam is not None:
            fa, flat = self._flat_adam, self.policy._flat
            slot = {id(p): sl for p, sl in zip(flat["params"], flat["slots"])}
            step = float(fa.step_count)
            if step > 0:
                for i, p in enumerate(params):
                    off, k = slot[id(p)]
                    state[i] = {"step": torch.tensor(step), "exp_avg": fa.m[off:off + k].view(p.shape).cpu().clone(),
                                "exp_avg_sq": fa.v[off:off + k].view(p.shape).cpu().clone()}
        else:
            sd = self.optimizer.state_dict()
            state = {i: {k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                     for i, st in sd["state"].items()}
        return {"state": state, "param_groups": [{"lr": cfg.learning_rate, "betas": (0.9, 0.999), "eps": 1e-5, "weight_decay": 0,
                                                  "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                                                  "params": list(range(len(params)))}]}

    def load_optimizer_state(self, opt: Optional[dict]) -> bool:
        """Restore Adam's moments from ``policy.optimizer.pth`` of a stable-baselines3 zip (``RecurrentPPO.load`` does on
        every curriculum resume, /root/reference/src/train/trainer.py:51-56).  State i belongs to the i-th parameter in
        stable-baselines3's order, which is this policy's ``parameters()`` order.  Returns False (and changes nothing)
        when the state is empty or its shapes do not fit."""
        if not opt or not opt.get("state"):
            return False
        params = list(self.policy.parameters())
        st = opt["state"]
        if sorted(st) != list(range(len(params))) or any(tuple(st[i]["exp_avg"].shape) != tuple(p.shape) for i, p in enumerate(params)):
            return False
        step = float(st[0]["step"]) if "step" in st[0] else 0.0
        if self._flat_adam is not None:
            fa, flat = self._flat_adam, self.policy._flat
            slot = {id(p): sl for p, sl in zip(flat["params"], flat["slots"])}
            for i, p in enumerate(params):
                off, k = slot[id(p)]
                fa.m[off:off + k].copy_(st[i]["exp_avg"].reshape(-1).to(fa.m))
                fa.v[off:off + k].copy_(st[i]["exp_avg_sq"].reshape(-1).to(fa.v))
            fa.set_step_count(step)
        else:
            dev = self.device
            for i, p in enumerate(params):
                self.optimizer.state[p] = {"step": torch.tensor(step, dtype=torch.float32, device=dev if self.device.type == "cuda" else "cpu"),
                                           "exp_avg": st[i]["exp_avg"].to(p).clone(), "exp_avg_sq": st[i]["exp_avg_sq"].to(p).clone()}
        return True

    def save(self, path: str) -> None:
        """``model.save(path)`` of the reference's callbacks (src/metrics/custom_callbacks.py:59): a
        stable-baselines3-format zip (rl/sb3_zip.save_sb3_zip)."""
        from .sb3_zip import save_sb3_zip
        cfg = self.cfg
        hyper = {k: getattr(cfg, k) for k in ("n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "ent_coef", "vf_coef",
                                              "max_grad_norm", "learning_rate", "clip_range", "normalize_advantage")}
        last = self._last_obs.detach().cpu().numpy() if self._last_obs is not None else None
        orig = getattr(self.env, "old_obs", None)
        save_sb3_zip(path, self.policy, hyper, n_envs=self.env.num_envs, num_timesteps=self.num_timesteps, n_updates=self.n_updates,
                     last_obs=last, last_original_obs=orig.detach().cpu().numpy() if orig is not None else None,
                     last_episode_starts=self._last_starts.detach().cpu().numpy() > 0.5,
                     optimizer_state=self._optimizer_state_dict(), clip_obs=float(getattr(self.env, "clip_obs", 10.0)))

    def set_parameters(self, path: str) -> None:
        """Load policy weights from a stable-baselines3 zip (``RecurrentPPO.load`` of src/train/trainer.py:51-56)."""
        from .sb3_zip import read_zip
        _, sd, _ = read_zip(path)
        self.policy.load_state_dict(sd, strict=True)        # in place: the flat-vector views stay valid
        if self._fused is not None:
            self._fused.refresh_shadow()

    def learn(self, total_timesteps: int, callback: Optional[Callable[["PPO"], None]] = None, log=None):
        """SB3's loop: collect a rollout (callbacks see every env step of it: `callback(self)` runs at the END of the rollout, BEFORE the
        update, with `self.n_calls` = vec-env steps so far — the policy a mid-rollout `_on_step` would have seen), train, log.
        Logged: time/fps, time/total_timesteps, rollout/ep_rew_mean and rollout/ep_len_mean (last 100 finished episodes, raw returns:
        SB3's Monitor / ep_info_buffer semantics), train/*."""
        t_start = time.time()
        while self.num_timesteps < total_timesteps:
            self.collect_rollouts()
            if callback is not None:
                callback(self)
            stats = self.train()
            if log is not None and self.rank == 0:
                fps = self.num_timesteps / max(1e-9, time.time() - t_start)
                log({"time/fps": fps, "time/total_timesteps": self.num_timesteps, **self.episode_stats(),
                     **{"train/" + k: v for k, v in stats.items()}})
        return self

    @property
    def n_calls(self) -> int:
        """vec-env steps taken so far (SB3's BaseCallback.n_calls: one per env.step of the vectorised env, whatever its width)"""
        return self.num_timesteps // max(1, self.env.num_envs * self.world)
