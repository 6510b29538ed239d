"""Hand-derived minibatch step of recurrent PPO (LSTM actor + LSTM critic -> MLP trunks -> heads).

The policy class the reference trains (``RecurrentPPO`` with ``lstm_hidden_size`` + ``net_arch``:
/root/reference/src/train/trainer.py:49-71, /root/reference/src/main_reorient.py:53-71).  Same loss and the same
minibatches as ``PPO._rec_forward_backward`` (whole rollouts of a subset of envs, LSTM state of the rollout start,
state zeroed where an episode starts — sb3-contrib ``_process_sequence``), without autograd and
without autocast: under those two the step is ~480 launches, two thirds of them casts, gradient adds and copies
around ~150 that do the work.  Here

* the minibatch is gathered by ``myo_ppo_gather`` (bf16 observations, advantage moments), one GEMM projects the
  inputs of both LSTMs for all time steps;
* the recurrence over ALL time steps is ONE launch per direction, ``myo_lstm_seq_fwd`` / ``myo_lstm_seq_bwd`` (csrc/myo_lstm_seq.h:
  a workgroup owns 16 sequences, the state goes from step to step through LDS and registers; hidden sizes 128 / 256, minibatches
  of a multiple of 16 sequences); else a time step is one launch per direction, ``myo_lstm_step_fwd`` / ``myo_lstm_step_bwd``
  (csrc/myo_lstm_step.h: recurrent product on the matrix cores with the cell arithmetic as its epilogue; hidden sizes 32 / 64 /
  128 / 256); else the batched recurrent GEMM + ``myo_lstm_cell_fwd`` / ``myo_lstm_cell_bwd`` (as ``rl/policy.py:_LstmSeq``);
* trunks, heads, loss and their backward pass are ``FusedPPOStep._merged_core`` on the LSTM outputs, which also
  returns the gradient entering the LSTMs;
* LSTM weight gradients are per-time-step batched GEMMs reduced in fp32 by ``myo_splitk_reduce`` (the bias
  gradient rides along as a ones column of the input), written straight into the flat gradient vector.

Every gradient slot is overwritten (no zeroing pass); clip + Adam follow as ``FlatAdam.step``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from .fused_mlp import FusedPPOStep


def lstm_seq_weights(whh, rs=1):
    """W_hh bf16 [G, 4H, H] -> (w_frag, wt_frag), the fragment-major layouts ``myo_lstm_seq_fwd`` / ``_bwd`` read (include/myobatch.h):
    the 64 x 16 bytes a wave loads for one matrix-core operand are one contiguous KB in lane order (lane = 16 (k-quarter) + row);
    a workgroup has eight waves, wave w owns UT = H / 128 tiles of 16 units and row i = 4 a + b of its tile ut is unit
    16 UT w + 4 UT a + CL (b // (4 / rs)) + (4 / rs) ut + b % (4 / rs), CL = 4 UT / rs: the cells of a lane — of each of the rs lane copies
    of a row when a workgroup owns 16 / rs rows — are consecutive units (csrc/myo_lstm_seq.h)."""
    G, H4, H = whh.shape
    UT, bpu = H // 128, 4 // rs
    # a unit's index digits, most significant first: w (8), a (4), s = b // bpu (rs), ut (UT), blo = b % bpu (bpu)
    #            [g][q][w][a][s][ut][blo][kk][lk][j]                        -> [g][w][kk][q][ut][lk][lr = (a, s, blo)][j]
    w_frag = whh.view(G, 4, 8, 4, rs, UT, bpu, H // 32, 4, 8).permute(0, 2, 7, 1, 5, 8, 3, 4, 6, 9).contiguous()
    #             [g][kk][lk][j][w][a][s][ut][blo]                          -> [g][w][kk][ut][lk][lr = (a, s, blo)][j]
    wt_frag = whh.view(G, H4 // 32, 4, 8, 8, 4, rs, UT, bpu).permute(0, 4, 1, 7, 2, 5, 6, 8, 3).contiguous()
    return w_frag, wt_frag


def lstm_seq_rows(x_tm, N, H, gates=1, rs=1):
    """A tile-major array of the sequence kernels (c_new, ws, cm from slot 1 on: the bytes of [..., G, N, gates * H], laid out
    [(g, row tile of 16 / rs)][wave][gate][lane = (lk, lane copy s, tile row)][CL]) as row-major [..., G, N, gates * H] (tests, diagnostics)."""
    rows, cl = 16 // rs, 4 * (H // 128) // rs
    lead = x_tm.shape[:-2]
    v = x_tm.reshape(*lead, N // rows, 8, gates, 4, rs, rows, cl)        # [rt][w][q][lk][s][row][CL]
    n = len(lead)
    v = v.permute(*range(n), n, n + 5, n + 2, n + 1, n + 3, n + 4, n + 6)   # [rt][row][q][w][lk][s][CL]: unit = 16 UT w + 4 UT lk + CL s + e
    return v.reshape(*lead, N, gates * H)


def lstm_seq_row_split(H, G, m, n_cu=256):
    """Rows per workgroup of the sequence kernels = 16 / rs: the largest split that does not leave the launch with more workgroups
    than CUs (the cell arithmetic of a step is per workgroup; MYO_LSTM_SEQ_RS overrides)."""
    forced = os.environ.get("MYO_LSTM_SEQ_RS")
    if forced:
        return int(forced)
    rs = 1
    for cand in ((2, 4) if H == 256 else (2,)):
        if G * m * cand // 16 <= n_cu:
            rs = cand
    return rs


class FusedRecurrentPPOStep(FusedPPOStep):
    """bf16 storage, fp32 accumulation; all shapes static (captured in a hipGraph by ``PPO._build_recurrent_graphs``)."""

    @classmethod
    def create(cls, policy, lib, clip_range, ent_coef, vf_coef):
        """None when the parameter layout has no stacked actor/critic views (odd sizes) — the caller keeps autograd."""
        if not policy.recurrent or getattr(policy, "_flat", None) is None:
            return None
        la, lc = policy.lstm_actor, policy.lstm_critic
        if lc is None or la.hidden_size != lc.hidden_size or la.hidden_size % 16:
            return None
        self = cls(policy, lib, clip_range, ent_coef, vf_coef)
        if self.merged is None:
            return None
        self.lstm = self._lstm_views(policy._flat, self.half[0])
        # one launch per time step and direction where the hidden size has a fused kernel (MYO_LSTM_TWO_KERNELS=1: GEMM + cell kernel)
        self.step_kernels = bool(lib.L.myo_lstm_step_supported(la.hidden_size)) and os.environ.get("MYO_LSTM_TWO_KERNELS") != "1"
        # ... and ALL time steps of a minibatch in one launch per direction where the hidden size has a sequence kernel
        # (csrc/myo_lstm_seq.h; MYO_LSTM_SEQ=0: one launch per time step)
        self.seq_kernels = self.step_kernels and bool(lib.L.myo_lstm_seq_supported(la.hidden_size)) and os.environ.get("MYO_LSTM_SEQ") != "0"
        # the CELL state travels from step to step in float32 (c_prev32 / c0_32 of the kernels: what nn.LSTM carries; h stays the bf16
        # MFMA operand); MYO_LSTM_C32=0 is the A/B switch back to a bf16 round trip per step
        self.c32 = os.environ.get("MYO_LSTM_C32") != "0"
        return self if self.lstm is not None else None

    def _lstm_views(self, flat, hflat):
        """[2, ...] views (actor, critic) over adjacent slots of the four LSTM parameters: bf16 shadow + fp32 gradient."""
        la, lc = self.policy.lstm_actor, self.policy.lstm_critic
        slot = {id(p): sl for p, sl in zip(flat["params"], flat["slots"])}
        out = {}
        for key in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            pa, pb = getattr(la, key), getattr(lc, key)
            (oa, k), (ob, _) = slot[id(pa)], slot[id(pb)]
            if pa.shape != pb.shape or ob != oa + k:
                return None
            shape = (2,) + tuple(pa.shape)
            short = key.split("_")[0][0] + key.split("_")[1]          # wih whh bih bhh
            out[short + "h"] = hflat[oa:oa + 2 * k].view(shape)
            out[short + "g"] = flat["g"][oa:oa + 2 * k].view(shape)
        return out

    def _static(self, key, shape, dtype, fill=None):
        """Buffers that live across calls (a captured graph keeps their addresses)."""
        t = self._work.get((key, shape, dtype))
        if t is None:
            t = torch.zeros(shape, dtype=dtype, device=self.acc.device) if fill is None else \
                torch.full(shape, fill, dtype=dtype, device=self.acc.device)
            self._work[(key, shape, dtype)] = t
        return t

    @torch.no_grad()
    def run_sequences(self, obs_buf, act_buf, start_buf, logp_buf, adv, ret, h0, c0, idx):
        """One minibatch step on the whole-rollout sequences of envs `idx`.

        obs_buf [T,N,O], act_buf [T,N,A], start_buf / logp_buf / adv / ret [T,N] (float32), h0 / c0 [2,N,H] float32
        (actor, critic LSTM state at the rollout start), idx int64 [m].  Returns (policy loss, value loss) as views of
        the loss kernel's accumulator."""
        lib, L = self.lib, self.lstm
        if self.adam is not None:
            self.adam.presummed = 0
        T, N, O = obs_buf.shape
        m, A = idx.shape[0], self.A
        B, H = T * m, self.policy.lstm_actor.hidden_size
        dev, bf = obs_buf.device, torch.bfloat16
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        if not self.adam_syncs_shadow:
            self.refresh_shadow()
        # ---- gather: rows (t, idx[j]) of the [T*N] rollout arrays, t-major
        trow = self._work.get(("trow", T, N))
        if trow is None:
            trow = self._work[("trow", T, N)] = (torch.arange(T, device=dev) * N).view(T, 1)
        ridx = (trow + idx.view(1, m)).reshape(B)
        x = torch.empty((1, B, O), device=dev, dtype=bf)
        act, oldlp = torch.empty((B, A), device=dev), torch.empty(B, device=dev)
        adv_mb, ret_mb = torch.empty(B, device=dev), torch.empty(B, device=dev)
        work = self._workbuf("gather", 2 * ((B + 15) // 16))
        lib.check(lib.L.myo_ppo_gather(p(obs_buf), p(act_buf), p(logp_buf), p(adv), p(ret), p(ridx), B, O, A, p(x), 1, p(act), p(oldlp),
                                       p(adv_mb), p(ret_mb), None if self.external_adv_stats else p(self.stats), p(work), st))
        keep = torch.rsub(start_buf.reshape(T * N).index_select(0, ridx), 1.0).view(T, m)       # 0 where an episode starts
        G, H4 = 2, 4 * H
        hm = torch.empty((T + 1, G, m, H), dtype=bf, device=dev)         # masked state entering step t
        cm = torch.empty_like(hm)
        k0 = keep[0].view(1, m, 1)
        torch.mul(h0.index_select(1, idx), k0, out=hm[0])
        c0m = c0.index_select(1, idx).float() * k0                        # the masked cell state entering step 0, UNROUNDED: the forward
        cm[0].copy_(c0m)                                                  # pass carries c in float32 (the bf16 slots are the backward pass's)
        # ---- input projections of both LSTMs, all time steps: [B, O] x [O, 2*4H]
        bsum = (L["bihh"] + L["bhhh"]).view(G * H4)
        gx = torch.addmm(bsum, x[0], L["wihh"].view(G * H4, O).t()).view(T, m, G, H4)      # row (t, n): [actor 4H | critic 4H]
        cn = torch.empty((T, G, m, H), dtype=bf, device=dev)
        ws = torch.empty((T, G, m, H4), dtype=bf, device=dev)
        lat = torch.empty((G, T, m, H), dtype=bf, device=dev)            # LSTM outputs, net-major: the trunks' input
        dG = torch.empty((T, G, m, H4), dtype=bf, device=dev)
        dcm = torch.empty((2, G, m, H), dtype=bf, device=dev)            # ping-pong: gradient of cm[t]
        kp = lambda t: p(keep[t + 1]) if t + 1 < T else None
        if self.seq_kernels and m % 16 == 0:
            # the whole sequence = ONE launch per direction (csrc/myo_lstm_seq.h): a workgroup owns 16 sequences from the first step to the last
            rs = lstm_seq_row_split(H, G, m)
            w_frag, wt_frag = lstm_seq_weights(L["whhh"], rs)
            lib.check(lib.L.myo_lstm_seq_fwd(p(gx), m * G * H4, H4, G * H4, p(hm), p(cm), p(w_frag), p(keep), G, m, H, T, rs, p(lat), T * m * H,
                                             m * H, p(cn), p(ws), p(c0m) if self.c32 else None, st))
            pl, vl, dlat = self._merged_core(lat.view(G, B, H), act, oldlp, adv_mb, ret_mb, want_dx=True)
            lib.check(lib.L.myo_lstm_seq_bwd(p(dlat), T * m * H, m * H, p(wt_frag), p(keep), p(cm), p(cn), p(ws), G, m, H, T, rs, p(dG), st))
        elif self.step_kernels:
            # a time step = ONE launch per direction (csrc/myo_lstm_step.h): recurrent product on the matrix cores + cell epilogue;
            # reads gx where the projection GEMM left it and writes the outputs where the trunks read them (no transposes)
            c32 = (c0m, torch.empty_like(c0m)) if self.c32 else (None, None)      # the float32 cell state, ping-pong
            for t in range(T):
                lib.check(lib.L.myo_lstm_step_fwd(p(gx[t]), H4, G * H4, p(hm[t]), p(cm[t]), p(L["whhh"]), kp(t), G, m, H, p(lat[:, t]),
                                                  T * m * H, p(hm[t + 1]), p(cm[t + 1]), p(cn[t]), p(ws[t]), p(c32[t & 1]), p(c32[(t + 1) & 1]), st))
            pl, vl, dlat = self._merged_core(lat.view(G, B, H), act, oldlp, adv_mb, ret_mb, want_dx=True)
            dlat = dlat.view(G, T, m, H)
            wt = L["whhh"].transpose(1, 2).contiguous()                  # [G, H, 4H] = W_hh^T
            for t in range(T - 1, -1, -1):
                last = t == T - 1
                lib.check(lib.L.myo_lstm_step_bwd(p(dlat[:, t]), T * m * H, None if last else p(dG[t + 1]), None if last else p(dcm[(t + 1) & 1]),
                                                  p(wt), kp(t), p(cm[t]), p(cn[t]), p(ws[t]), G, m, H, p(dG[t]), p(dcm[t & 1]), st))
        else:
            gxs = gx.transpose(1, 2).contiguous()                        # [T, G, m, 4H]: a step's rows are r = g*m + n
            out = torch.empty((T, G, m, H), dtype=bf, device=dev)
            wt = L["whhh"].transpose(1, 2).contiguous()                  # [G, H, 4H] = W_hh^T
            for t in range(T):
                gh = torch.bmm(hm[t], wt)
                lib.check(lib.L.myo_lstm_cell_fwd(p(gxs[t]), p(gh), p(cm[t]), kp(t), G * m, m, H, 1,
                                                  p(out[t]), p(hm[t + 1]), p(cm[t + 1]), p(cn[t]), p(ws[t]), st))
            lat.copy_(out.transpose(0, 1))
            pl, vl, dlat = self._merged_core(lat.view(G, B, H), act, oldlp, adv_mb, ret_mb, want_dx=True)
            dout = dlat.view(G, T, m, H).transpose(0, 1).contiguous()
            whh = L["whhh"]                                              # [G, 4H, H]: dh_prev = dgates . W_hh
            dhm = dcn = None
            for t in range(T - 1, -1, -1):
                lib.check(lib.L.myo_lstm_cell_bwd(p(dout[t]), p(dhm), p(dcn), kp(t), p(cm[t]), p(cn[t]),
                                                  p(ws[t]), G * m, m, H, 1, p(dG[t]), p(dcm[t & 1]), st))
                if t > 0:
                    dhm, dcn = torch.bmm(dG[t], whh), dcm[t & 1]
        # ---- LSTM weight gradients: per-time-step partial products (batch = (t, g)), summed over t in fp32
        Op = (O + 1 + 7) // 8 * 8                                        # inputs + a ones column (bias gradient) + zero padding
        xe = self._static("xe", (T * G, m, Op), bf)
        if not self._work.get(("xe_init", T, m, Op)):
            xe[..., O] = 1.0
            self._work[("xe_init", T, m, Op)] = True
        xe.view(T, G, m, Op)[..., :O].copy_(x.view(T, 1, m, O).expand(T, G, m, O))
        dGt = dG.view(T * G, m, H4).transpose(1, 2)
        part_hh = torch.bmm(dGt, hm[:T].view(T * G, m, H))               # [T*G, 4H, H]
        self._reduce(part_hh, L["whhg"], 1, T)
        part_ih = torch.bmm(dGt, xe)                                     # [T*G, 4H, Op]
        gih = self._static("gih", (G, H4, Op), torch.float32)
        self._reduce(part_ih, gih, 1, T)
        L["wihg"].copy_(gih[..., :O])
        L["bihg"].copy_(gih[..., O])
        L["bhhg"].copy_(L["bihg"])
        return pl, vl
