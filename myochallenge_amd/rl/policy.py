"""Actor-critic policies with stable-baselines3 / sb3-contrib parameter naming.

The reference trains ``RecurrentPPO("MlpLstmPolicy", ...)`` (/root/reference/src/train/trainer.py:57-64)
with ``policy_kwargs`` such as ``lstm_hidden_size=256, net_arch=[dict(pi=[256,256], vf=[256,256])],
enable_critic_lstm=True, ortho_init=False, activation_fn=ReLU, log_std_init=-2``
(trained_models/curriculum_steps_complete_baoding_winner/01_rsi_static/main.py:178-200) and ships
one checkpoint whose ``policy.pth`` has LSTM-128 actor/critic and no MLP
(trained_models/phase_1/phase1_final.zip [ART]).  State-dict keys below are exactly SB3's, so
``policy.pth`` loads with ``load_state_dict`` unchanged (tests/test_checkpoint.py).

``lstm_hidden_size=None`` gives the plain MLP policy of BASELINE.json's configs ("PPO MLP[256,256]").
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn


def _mlp(inp: int, sizes: Sequence[int], act) -> nn.Sequential:
    layers = []
    for h in sizes:
        layers += [nn.Linear(inp, h), act()]
        inp = h
    return nn.Sequential(*layers)


class _LinearGemmBias(torch.autograd.Function):
    """y = x @ w.T + b whose bias gradient is a [1,B] x [B,out] GEMM instead of a column reduction.

    Why: inside a replayed hipGraph, ATen's multi-block column reduction (many outputs, many rows: semaphore-based
    global reduce) returns wrong sums on this ROCm stack once the graph's private pool has been reused
    (tools/dev/gpu_reduce_graph.py reproduces it: every replay after the first is wrong) — it put a stray 1e32
    into one LSTM bias gradient.  GEMMs, row reductions and single-output reductions replay correctly."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        ones = torch.ones((1, g.shape[0]), dtype=g.dtype, device=g.device)
        return g @ w, g.t() @ x, (ones @ g).squeeze(0)


def _cell_fwd(gx, gh, c):
    """LSTM pointwise cell on pre-activations gx + gh ([R,4H], gate order i,f,g,o) -> (h, c_new, workspace)."""
    if gx.is_cuda and hasattr(torch.ops.aten, "_thnn_fused_lstm_cell"):
        return torch.ops.aten._thnn_fused_lstm_cell(gx, gh, c, None, None)
    i, f, g, o = (gx + gh).chunk(4, dim=-1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    cn = f * c + i * g
    return o * torch.tanh(cn), cn, torch.cat([i, f, g, o], -1)


def _cell_bwd(dh, dc, c_prev, c_new, ws):
    """Backward of _cell_fwd: -> (d pre-activations [R,4H], d c_prev)."""
    if dh.is_cuda and hasattr(torch.ops.aten, "_thnn_fused_lstm_cell_backward_impl"):
        dg, dcp, _ = torch.ops.aten._thnn_fused_lstm_cell_backward_impl(dh, dc, c_prev, c_new, ws, False)
        return dg, dcp
    i, f, g, o = ws.chunk(4, dim=-1)
    tc = torch.tanh(c_new)
    dct = dc + dh * o * (1 - tc * tc)
    dg = torch.cat([dct * g * i * (1 - i), dct * c_prev * f * (1 - f), dct * i * (1 - g * g), dh * tc * o * (1 - o)], -1)
    return dg, dct * f


def _native_lib(t):
    """libmyobatch (HIP build) for CUDA tensors in float32 / bfloat16, else None."""
    if not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16):
        return None
    try:
        from .. import native
        lib = native.load()
    except Exception:
        return None
    return lib


class _LstmSeq(torch.autograd.Function):
    """G stacked one-layer LSTMs over a whole sequence with a hand-written backward.

    gx [T,G,N,4H] input projections (+ biases), wt [G,H,4H] = W_hh^T, h0/c0 [G,N,H], keep [T,1,N,1] (0 where an
    episode starts: state zeroed before that step).  Autograd's own BPTT of the step loop costs ~10 small
    kernels per step (select / stack backwards, gradient adds, a weight-gradient GEMM and its accumulation per
    step).  Here a time step is TWO launches in each direction on the GPU — the batched recurrent GEMM and one
    libmyobatch kernel (``myo_lstm_cell_fwd`` / ``myo_lstm_cell_bwd``: gates, cell, the next step's episode
    mask, and in the backward pass the gradient add) — and the recurrent weight gradient is ONE batched GEMM
    over all steps after the loop.  Without the HIP library (CPU) the same recurrences run on ATen ops."""

    @staticmethod
    def forward(ctx, gx, wt, h0, c0, keep):
        T, G, N, H4 = gx.shape
        H = H4 // 4
        lib = _native_lib(gx)
        ctx.native = lib is not None
        if lib is not None:
            import ctypes as C
            # The CELL (gates, cell state, what the backward pass keeps) is float32 whatever the GEMM dtype: c is the state an LSTM
            # carries through an episode — stock nn.LSTM carries it in float32 (/root/reference/src/main_reorient.py:53-71), and so
            # do the fused paths (csrc/myo_lstm_step.h / myo_lstm_seq.h: c_prev32) — while h is rounded to the GEMM dtype, its only use.
            gd, f32 = gx.dtype, torch.float32
            gx = gx.contiguous()
            kf = None if keep is None else keep.reshape(T, N).float().contiguous()
            hm = torch.empty((T + 1, G, N, H), dtype=gd, device=gx.device)           # masked h entering step t (the recurrent GEMM's input)
            cm = torch.empty((T + 1, G, N, H), dtype=f32, device=gx.device)
            out, cn = torch.empty((T, G, N, H), dtype=f32, device=gx.device), torch.empty((T, G, N, H), dtype=f32, device=gx.device)
            ws = torch.empty((T, G, N, H4), dtype=f32, device=gx.device)
            hmf = torch.empty((G, N, H), dtype=f32, device=gx.device)
            hm[0] = h0 if kf is None else h0 * keep[0]
            cm[0] = c0.float() if kf is None else c0.float() * keep[0].float()
            st = C.c_void_p(torch.cuda.current_stream(gx.device).cuda_stream)
            p = lambda t: C.c_void_p(t.data_ptr())
            wt = wt.contiguous()
            for t in range(T):
                gh = torch.bmm(hm[t], wt).float()
                gxt = gx[t].float()
                kn = p(kf[t + 1]) if (kf is not None and t + 1 < T) else None
                lib.check(lib.L.myo_lstm_cell_fwd(p(gxt), p(gh), p(cm[t]), kn, G * N, N, H, 0, p(out[t]), p(hmf), p(cm[t + 1]),
                                                  p(cn[t]), p(ws[t]), st))
                hm[t + 1].copy_(hmf)
            ctx.save_for_backward(wt, hm, cm, cn, ws, kf if kf is not None else wt.new_zeros(0))
            ctx.has_keep = kf is not None
            ctx.gemm_dtype = gd
            return out.to(gd), hm[T], cm[T]                # no mask after the last step: the final state itself (c in float32)
        h, c = h0, c0
        hs, cs, cn, wss, outs = [], [], [], [], []
        for t in range(T):
            if keep is not None:
                h, c = h * keep[t], c * keep[t]
            gh = torch.bmm(h, wt)
            h2, c2, ws = _cell_fwd(gx[t].reshape(G * N, H4), gh.reshape(G * N, H4), c.reshape(G * N, H))
            hs.append(h); cs.append(c); cn.append(c2); wss.append(ws)
            h, c = h2.view(G, N, H), c2.view(G, N, H)
            outs.append(h)
        ctx.save_for_backward(wt, torch.stack(hs, 0), torch.stack(cs, 0), torch.stack(cn, 0), torch.stack(wss, 0),
                              keep if keep is not None else wt.new_zeros(0))
        ctx.has_keep = keep is not None
        return torch.stack(outs, 0), h, c

    @staticmethod
    def backward(ctx, dout, dhT, dcT):
        wt, hs, cs, cn, wss, keep = ctx.saved_tensors
        wtt = wt.transpose(1, 2).contiguous()
        if ctx.native:
            import ctypes as C
            from .. import native
            lib = native.load()
            T, G, N, H = cn.shape
            gd, f32 = ctx.gemm_dtype, torch.float32
            dG = torch.empty((T, G, N, 4 * H), dtype=gd, device=cn.device)
            dGf = torch.empty((G, N, 4 * H), dtype=f32, device=cn.device)          # (the cell's gradients in float32, as its forward pass)
            dcm = torch.empty((2, G, N, H), dtype=f32, device=cn.device)           # ping-pong: gradient of cm[t]
            st = C.c_void_p(torch.cuda.current_stream(cn.device).cuda_stream)
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            dout = dout.float().contiguous()
            dhm = None if dhT is None else dhT.float().contiguous()      # gradient of hm[t+1] (the final state at t = T-1)
            dcn = None if dcT is None else dcT.float().contiguous()
            for t in range(T - 1, -1, -1):
                kn = p(keep[t + 1]) if (ctx.has_keep and t + 1 < T) else None
                lib.check(lib.L.myo_lstm_cell_bwd(p(dout[t]), p(dhm), p(dcn), kn, p(cs[t]), p(cn[t]), p(wss[t]), G * N, N, H, 0,
                                                  p(dGf), p(dcm[t & 1]), st))
                dG[t].copy_(dGf)
                dhm, dcn = torch.bmm(dG[t], wtt).float(), dcm[t & 1]
            if ctx.has_keep:
                k0 = keep[0].view(1, N, 1).to(dhm.dtype)
                dhm, dcn = dhm * k0, dcn * k0
            hm_in = hs[:T].transpose(0, 1).reshape(G, T * N, H)
            dwt = torch.bmm(hm_in.transpose(1, 2), dG.transpose(0, 1).reshape(G, T * N, 4 * H))
            return dG, dwt, dhm.to(gd), dcn.clone(), None
        T, G, N, H = hs.shape
        dh = dhT if dhT is not None else torch.zeros_like(hs[0])
        dc = dcT if dcT is not None else torch.zeros_like(hs[0])
        dG = torch.empty((T, G, N, 4 * H), dtype=hs.dtype, device=hs.device)
        for t in range(T - 1, -1, -1):
            dg, dcp = _cell_bwd((dout[t] + dh).reshape(G * N, H).contiguous(), dc.reshape(G * N, H).contiguous(),
                                cs[t].reshape(G * N, H), cn[t], wss[t])
            dG[t] = dg.view(G, N, 4 * H)
            dh, dc = torch.bmm(dG[t], wtt), dcp.view(G, N, H)
            if ctx.has_keep:
                dh, dc = dh * keep[t], dc * keep[t]
        # dW_hh^T[g] = sum_t h_{t-1}^T dgates_t : one batched GEMM over the flattened (t, n) rows
        hm = hs.transpose(0, 1).reshape(G, T * N, H)
        dgm = dG.transpose(0, 1).reshape(G, T * N, 4 * H)
        dwt = torch.bmm(hm.transpose(1, 2), dgm)
        return dG, dwt, dh, dc, None


def _linear(x, w, b):
    """F.linear for 2-D x; on the GPU with gradients enabled it goes through _LinearGemmBias (autocast-aware)."""
    if x.is_cuda and torch.is_grad_enabled() and (w.requires_grad or x.requires_grad):
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            x, w, b = x.to(dt), w.to(dt), b.to(dt)
        return _LinearGemmBias.apply(x, w, b)
    return torch.nn.functional.linear(x, w, b)


def _apply_net(net, x):
    """nn.Sequential / nn.Linear forward with every Linear routed through _linear (any leading dims)."""
    lead = x.shape[:-1]
    x = x.reshape(-1, x.shape[-1])
    for layer in (net if isinstance(net, nn.Sequential) else [net]):
        x = _linear(x, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(x)
    return x.reshape(*lead, x.shape[-1])


class MlpExtractor(nn.Module):
    def __init__(self, inp: int, pi: Sequence[int], vf: Sequence[int], act):
        super().__init__()
        self.policy_net = _mlp(inp, pi, act)
        self.value_net = _mlp(inp, vf, act)
        self.latent_dim_pi = pi[-1] if pi else inp
        self.latent_dim_vf = vf[-1] if vf else inp


class ActorCriticPolicy(nn.Module):
    def __init__(self, obs_dim: int, act_dim: int, pi: Sequence[int] = (256, 256), vf: Sequence[int] = (256, 256),
                 lstm_hidden_size: Optional[int] = None, enable_critic_lstm: bool = True,
                 log_std_init: float = -2.0, activation_fn=nn.ReLU, use_sde: bool = False):
        """use_sde: generalised state-dependent exploration (stable-baselines3's StateDependentNoiseDistribution with its
        defaults full_std=True, use_expln=False, squash_output=False, learn_features=False; SURVEY.md Appendix C.4;
        /root/reference/docs/summary.md:100, `use_sde=True` in the archived curriculum scripts): ``log_std`` is a
        [latent_pi, act] matrix, the exploration noise of env n is latent_pi(s) . W_n with W_n ~ N(0, exp(log_std))
        drawn by ``reset_noise`` once per rollout (sde_sample_freq = -1), and the action distribution is
        Normal(mean, sqrt(latent_pi^2 . exp(log_std)^2 + 1e-6))."""
        super().__init__()
        self.use_sde = bool(use_sde)
        self.exploration_mat = None          # [n_envs, latent_pi, act], set by reset_noise
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.pi_arch, self.vf_arch = list(pi), list(vf)
        self.lstm_hidden_size, self.enable_critic_lstm = lstm_hidden_size, enable_critic_lstm
        self.recurrent = lstm_hidden_size is not None
        if self.recurrent and not enable_critic_lstm:
            raise NotImplementedError("shared / linear critic variants are not used by the reference")
        feat = lstm_hidden_size if self.recurrent else obs_dim
        self.hidden = lstm_hidden_size
        # Registration order = stable-baselines3's (ActorCriticPolicy._build, then RecurrentActorCriticPolicy adds its
        # LSTMs): parameters() yields log_std, mlp_extractor.*, action_net.*, value_net.*, lstm_actor.*, lstm_critic.*,
        # and policy.optimizer.pth of an SB3 zip indexes its Adam state by that order (state[1] of phase1_final.zip is
        # action_net.weight (39,128), state[5] lstm_actor.weight_ih_l0 (512,86)).
        self.mlp_extractor = MlpExtractor(feat, list(pi), list(vf), activation_fn)
        self.action_net = nn.Linear(self.mlp_extractor.latent_dim_pi, act_dim)
        self.value_net = nn.Linear(self.mlp_extractor.latent_dim_vf, 1)
        if self.recurrent:
            self.lstm_actor = nn.LSTM(obs_dim, lstm_hidden_size, num_layers=1)
            self.lstm_critic = nn.LSTM(obs_dim, lstm_hidden_size, num_layers=1)
        shape = (self.mlp_extractor.latent_dim_pi, act_dim) if self.use_sde else (act_dim,)
        self.log_std = nn.Parameter(torch.full(shape, float(log_std_init)))

    # ---- gSDE
    SDE_EPS = 1e-6

    @torch.no_grad()
    def reset_noise(self, n_envs: int, generator: Optional[torch.Generator] = None, out: Optional[torch.Tensor] = None) -> None:
        """One exploration matrix per env, W ~ N(0, exp(log_std)) (SB3 ``sample_weights``; called at the start of every
        rollout for sde_sample_freq = -1).  out: a caller-owned [n_envs, latent, act] buffer to draw into — PPO's captured rollout
        graph reads ITS buffer by address, so the training matrix must never move (an evaluation on another batch size
        between two rollouts used to replace the tensor and leave the graph reading freed memory)."""
        if not self.use_sde:
            return
        std = torch.exp(self.log_std.float())
        eps = torch.randn((n_envs,) + tuple(std.shape), device=std.device, generator=generator)
        if out is not None:
            out.copy_(eps * std)
            self.exploration_mat = out
        elif self.exploration_mat is not None and self.exploration_mat.shape == eps.shape:
            self.exploration_mat.copy_(eps * std)          # in place: a captured rollout graph keeps reading this tensor
        else:
            self.exploration_mat = eps * std

    def _sde_std(self, latent_pi):
        """Per-sample action std of the state-dependent distribution."""
        var = (latent_pi.float() ** 2) @ (torch.exp(self.log_std.float()) ** 2)
        return torch.sqrt(var + self.SDE_EPS)

    # ---- recurrent helpers
    def initial_state(self, n: int, device) -> Optional[Tuple[torch.Tensor, ...]]:
        if not self.recurrent:
            return None
        z = lambda: torch.zeros((1, n, self.hidden), device=device)
        return (z(), z(), z(), z())   # (h_pi, c_pi, h_vf, c_vf)

    @staticmethod
    def _lstm_cell_steps(lstm: nn.LSTM, x, h, c, starts):
        """One-layer LSTM written out as GEMMs + a fused gate kernel (PyTorch gate order i, f, g, o) — the GPU
        path: the library RNN (MIOpen) compiles its kernels on first use, minutes on a fresh machine.  The
        input projection of the whole sequence is ONE GEMM; per step there is the recurrent GEMM and ATen's
        fused pointwise cell (forward and backward are one kernel each).  No host sync on `starts`."""
        w_ih, w_hh, b_ih, b_hh = lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0
        st_dtype = h.dtype
        h, c = h[0], c[0]
        T, N = x.shape[0], x.shape[1]
        gx = _linear(x.reshape(T * N, -1), w_ih, b_ih + b_hh).view(T, N, -1)
        keep = None if starts is None else (1.0 - starts.to(gx.dtype)).unsqueeze(-1)
        h, c = h.to(gx.dtype), c.float()               # (the cell state stays float32 under autocast: see _LstmSeq)
        fused = x.is_cuda and hasattr(torch.ops.aten, "_thnn_fused_lstm_cell")
        wt = w_hh.t().to(gx.dtype)
        outs = []
        for t in range(T):
            if keep is not None:
                h, c = h * keep[t], c * keep[t].float()
            gh = (h @ wt).float()
            if fused:
                h, c, _ = torch.ops.aten._thnn_fused_lstm_cell(gx[t].float(), gh, c, None, None)
            else:
                i, f, g, o = (gx[t].float() + gh).chunk(4, dim=-1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = torch.sigmoid(o) * torch.tanh(c)
            h = h.to(gx.dtype)
            outs.append(h)
        return torch.stack(outs, 0), h.unsqueeze(0).to(st_dtype), c.unsqueeze(0).to(st_dtype)

    @staticmethod
    def _lstm_pair_steps(la: nn.LSTM, lc: nn.LSTM, x, state, starts):
        """Actor and critic LSTMs (same shapes, same input) advanced together: ONE input-projection GEMM for
        both networks and the whole sequence, then per step one batched recurrent GEMM and one fused cell
        kernel over the stacked [2N] rows — half the launches of two `_lstm_cell_steps` calls, same math."""
        ha, ca, hc, cc = state
        st_dtype = ha.dtype
        T, N, H = x.shape[0], x.shape[1], la.hidden_size
        w_ih = torch.cat([la.weight_ih_l0, lc.weight_ih_l0], 0)                       # [8H, F]
        b = torch.cat([la.bias_ih_l0 + la.bias_hh_l0, lc.bias_ih_l0 + lc.bias_hh_l0], 0)
        gx = _linear(x.reshape(T * N, -1), w_ih, b)                                   # [T*N, 8H]
        gx = gx.view(T, N, 2, 4 * H).transpose(1, 2).contiguous()                     # [T, 2, N, 4H]
        wt = torch.stack([la.weight_hh_l0.t(), lc.weight_hh_l0.t()], 0)               # [2, H, 4H]
        h = torch.stack([ha[0], hc[0]], 0).to(gx.dtype)                               # [2, N, H]
        c = torch.stack([ca[0], cc[0]], 0).float()                                    # (the cell state stays float32 under autocast: see _LstmSeq)
        keep = None if starts is None else (1.0 - starts.to(gx.dtype)).view(T, 1, N, 1)
        fused = x.is_cuda and hasattr(torch.ops.aten, "_thnn_fused_lstm_cell")
        wt = wt.to(gx.dtype)
        fin = lambda z, k: z[k].unsqueeze(0).to(st_dtype)
        if torch.is_grad_enabled() and gx.requires_grad and T > 1:      # training: hand-written BPTT
            out, h, c = _LstmSeq.apply(gx, wt, h, c, keep)
            return out[:, 0], out[:, 1], (fin(h, 0), fin(c, 0), fin(h, 1), fin(c, 1))
        outs = []
        for t in range(T):
            if keep is not None:
                h, c = h * keep[t], c * keep[t].float()
            gh = torch.bmm(h, wt).float()                                             # [2, N, 4H]
            if fused:
                h2, c2, _ = torch.ops.aten._thnn_fused_lstm_cell(gx[t].float().reshape(2 * N, 4 * H), gh.reshape(2 * N, 4 * H),
                                                                 c.reshape(2 * N, H), None, None)
                h, c = h2.view(2, N, H), c2.view(2, N, H)
            else:
                i, f, g, o = (gx[t].float() + gh).chunk(4, dim=-1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h = torch.sigmoid(o) * torch.tanh(c)
            h = h.to(gx.dtype)
            outs.append(h)
        out = torch.stack(outs, 0)                                                     # [T, 2, N, H]
        return out[:, 0], out[:, 1], (fin(h, 0), fin(c, 0), fin(h, 1), fin(c, 1))

    @staticmethod
    def _lstm_seq(lstm: nn.LSTM, x, h, c, starts):
        """x [T,N,F]; starts [T,N] (1 = episode start: state zeroed before that step) — sb3-contrib
        ``_process_sequence`` [3P-RECALL, SURVEY.md C.3]."""
        if x.is_cuda:
            return ActorCriticPolicy._lstm_cell_steps(lstm, x, h, c, starts)
        if starts is None or not bool(starts.any()):
            out, (h, c) = lstm(x, (h, c))
            return out, h, c
        outs = []
        for t in range(x.shape[0]):
            keep = (1.0 - starts[t].to(x.dtype)).view(1, -1, 1)
            o, (h, c) = lstm(x[t:t + 1], (h * keep, c * keep))
            outs.append(o)
        return torch.cat(outs, 0), h, c

    def _latents(self, obs, state, starts):
        """obs [T,N,F] (recurrent) or [B,F]."""
        if self.recurrent:
            if obs.is_cuda and self.lstm_critic is not None and self.lstm_critic.hidden_size == self.lstm_actor.hidden_size:
                lp, lv, state = self._lstm_pair_steps(self.lstm_actor, self.lstm_critic, obs, state, starts)
            else:
                hp, cp, hv, cv = state
                lp, hp, cp = self._lstm_seq(self.lstm_actor, obs, hp, cp, starts)
                lv, hv, cv = self._lstm_seq(self.lstm_critic, obs, hv, cv, starts)
                state = (hp, cp, hv, cv)
        else:
            lp = lv = obs
        return _apply_net(self.mlp_extractor.policy_net, lp), _apply_net(self.mlp_extractor.value_net, lv), state

    def _dist(self, latent_pi):
        mean = _apply_net(self.action_net, latent_pi).float()
        return mean, self.log_std.float()

    @staticmethod
    def log_prob(actions, mean, log_std):
        var = torch.exp(2 * log_std)
        return (-0.5 * ((actions - mean) ** 2) / var - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)

    def entropy(self, latent_pi=None):
        """DiagGaussian: one number (state independent).  gSDE: per sample, from the state-dependent std."""
        if self.use_sde:
            return (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(self._sde_std(latent_pi.detach()))).sum(-1)
        return (0.5 + 0.5 * math.log(2 * math.pi) + self.log_std.float()).sum()

    # ---- rollout: one step for all envs
    @torch.no_grad()
    def act(self, obs, state=None, episode_starts=None, deterministic=False):
        x = obs.unsqueeze(0) if self.recurrent else obs
        st = episode_starts.unsqueeze(0) if (self.recurrent and episode_starts is not None) else None
        lp, lv, state = self._latents(x, state, st)
        if self.recurrent:
            lp, lv = lp[0], lv[0]
        mean, log_std = self._dist(lp)
        # exploration noise: N(0, 1) from torch's generator, or from `noise_fn(mean)` when one is installed (tests of
        # the N-rank == 1-rank equivalence draw ONE global noise tensor and give every rank its rows)
        values = _apply_net(self.value_net, lv).float().squeeze(-1)
        if self.use_sde:
            lat = lp.float()
            Wm = self.exploration_mat
            if Wm is None or Wm.shape[0] != lat.shape[0]:
                # another batch size than the one the matrix was drawn for (an evaluation between rollouts): a temporary of its own —
                # the training matrix stays where it is; a deterministic call needs no noise at all
                std = torch.exp(self.log_std.float())
                Wm = torch.zeros((lat.shape[0],) + tuple(std.shape), device=std.device) if deterministic else \
                    torch.randn((lat.shape[0],) + tuple(std.shape), device=std.device) * std
                if self.exploration_mat is None:
                    self.exploration_mat = Wm
            lib = _native_lib(lat)
            if lib is not None and lat.dtype == torch.float32 and self.SDE_EPS == 1e-6:
                # GPU rollout: noise, sigma and log pi of all envs in ONE launch (myo_rollout_sample_sde) instead of
                # bmm + the matmul of _sde_std + the elementwise log_prob chain
                import ctypes as C
                mu, lat = mean.float().contiguous(), lat.contiguous()
                W, ls = Wm.contiguous(), self.log_std.detach().float().contiguous()
                actions, clipped, logp = torch.empty_like(mu), torch.empty_like(mu), mu.new_empty(mu.shape[0])
                p = lambda t: C.c_void_p(t.data_ptr())
                lib.check(lib.L.myo_rollout_sample_sde(p(mu), p(lat), p(W), p(ls), mu.shape[0], lat.shape[1], mu.shape[1],
                                                       p(actions), p(clipped), p(logp), int(bool(deterministic)),
                                                       C.c_void_p(torch.cuda.current_stream(mu.device).cuda_stream)))
                return actions, values, logp, state
            noise = torch.bmm(lat.unsqueeze(1), Wm).squeeze(1)       # latent_pi(s_n) . W_n
            actions = mean if deterministic else mean + noise
            return actions, values, self.log_prob(actions, mean, torch.log(self._sde_std(lat))), state
        noise = torch.randn_like(mean) if getattr(self, "noise_fn", None) is None else self.noise_fn(mean)
        actions = mean if deterministic else mean + torch.exp(log_std) * noise
        return actions, values, self.log_prob(actions, mean, log_std), state

    @torch.no_grad()
    def predict_values(self, obs, state=None, episode_starts=None):
        """Critic only (the actor LSTM / trunk are not evaluated)."""
        lv = obs
        if self.recurrent:
            st = episode_starts.unsqueeze(0) if episode_starts is not None else None
            lv = self._lstm_seq(self.lstm_critic, obs.unsqueeze(0), state[2], state[3], st)[0][0]
        return _apply_net(self.value_net, _apply_net(self.mlp_extractor.value_net, lv)).float().squeeze(-1)

    # ---- training: evaluate stored actions
    def evaluate_actions(self, obs, actions, state=None, episode_starts=None):
        """MLP: obs [B,F], actions [B,A].  Recurrent: obs [T,N,F], actions [T,N,A], state at t=0."""
        lp, lv, _ = self._latents(obs, state, episode_starts)
        mean, log_std = self._dist(lp)
        values = _apply_net(self.value_net, lv).float().squeeze(-1)
        if self.use_sde:        # learn_features=False: the exploration features carry no gradient into the trunk
            lat = lp.detach()
            return values, self.log_prob(actions, mean, torch.log(self._sde_std(lat))), self.entropy(lat)
        return values, self.log_prob(actions, mean, log_std), self.entropy()

    def predict(self, observation, state=None, episode_start=None, deterministic=False):
        """SB3-style predict on numpy or tensors (used by evaluation loops such as
        /root/reference/src/main_eval.py:86-118)."""
        import numpy as np
        dev = self.log_std.device
        obs = torch.as_tensor(observation, dtype=torch.float32, device=dev)
        if obs.dim() == 1:
            obs = obs.unsqueeze(0)
        if self.recurrent and state is None:
            state = self.initial_state(obs.shape[0], dev)
        starts = None if episode_start is None else torch.as_tensor(np.asarray(episode_start), dtype=torch.float32, device=dev)
        a, _, _, state = self.act(obs, state, starts, deterministic)
        a = torch.clamp(a, -1.0, 1.0)
        return (a.cpu().numpy() if isinstance(observation, np.ndarray) else a), state
