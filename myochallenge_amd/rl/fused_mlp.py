"""Hand-derived forward/backward of the PPO minibatch loss for the MLP actor-critic.

Same loss as ``PPO._loss`` (clipped surrogate + vf_coef * MSE - ent_coef * entropy, per-minibatch
advantage normalisation; SB3 semantics) without autograd:

* bf16 GEMMs on contiguous operands, weights cast once per step with one multi-tensor copy;
* weight gradients dW = dY' X (16k-long reduction, 256x256 output) as a split-K ``bmm`` —
  hipBLASLt otherwise runs them on 16 workgroups of a 256-CU chip (94 us each, 22 % of the update);
* the whole elementwise part (log-prob, ratio, clip, value loss and their gradients) in ONE HIP
  kernel of libmyobatch (``myo_ppo_loss_grad``) instead of ~40 torch launches.

``ppo_mlp_step_grads`` is the pure-torch statement of the same math (CPU tests compare it with
autograd); ``FusedPPOStep`` is the GPU path that PPO captures into a hipGraph.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Tuple

import torch


def _layers(seq) -> List[torch.nn.Linear]:
    return [m for m in seq if isinstance(m, torch.nn.Linear)]


def _nets(policy):
    return {"pi": _layers(policy.mlp_extractor.policy_net) + [policy.action_net],
            "vf": _layers(policy.mlp_extractor.value_net) + [policy.value_net]}


def _splitk_wgrad(dy: torch.Tensor, x: torch.Tensor, split: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """dW = dy' x, db = dy' 1  with x [B,in], dy [B,out]; split-K over the batch."""
    B = x.shape[0]
    if split > 1 and B % split == 0:
        gw = torch.bmm(dy.view(split, B // split, -1).transpose(1, 2), x.view(split, B // split, -1)).sum(0, dtype=torch.float32)
    else:
        gw = (dy.t() @ x).float()
    return gw, dy.sum(0, dtype=torch.float32)


@torch.no_grad()
def ppo_mlp_step_grads(policy, obs, actions, old_logp, adv, returns, clip_range, ent_coef, vf_coef,
                       normalize_advantage=True, bf16=True, split_k=32):
    """Reference (pure torch) version: writes d(loss)/d(param) into ``p.grad``; returns (pl, vl)."""
    assert not policy.recurrent
    cd = torch.bfloat16 if (bf16 and obs.is_cuda) else torch.float32
    B = obs.shape[0]
    x0 = obs.to(cd)
    nets = _nets(policy)
    acts, outs = {}, {}
    for name, layers in nets.items():
        h = x0
        saved = [h]
        for li, lin in enumerate(layers):
            h = torch.addmm(lin.bias.to(cd), h, lin.weight.to(cd).t())
            if li < len(layers) - 1:
                h = torch.relu_(h)
                saved.append(h)
        acts[name], outs[name] = saved, h.float()
    mean, values = outs["pi"], outs["vf"].squeeze(-1)
    log_std = policy.log_std.float()
    inv_std = torch.exp(-log_std)
    z = (actions - mean) * inv_std
    logp = (-0.5 * z * z - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
    if normalize_advantage and B > 1:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    ratio = torch.exp(logp - old_logp)
    s1 = adv * ratio
    s2 = adv * torch.clamp(ratio, 1 - clip_range, 1 + clip_range)
    pl = -torch.min(s1, s2).mean()
    vl = torch.mean((values - returns) ** 2)
    inside = (ratio > 1 - clip_range) & (ratio < 1 + clip_range)
    dlogp = -(adv * ratio) * torch.where(s1 <= s2, torch.ones_like(ratio), inside.to(ratio.dtype)) / B
    dmean = dlogp.unsqueeze(-1) * z * inv_std
    dlogstd = (dlogp.unsqueeze(-1) * (z * z - 1)).sum(0) - ent_coef
    dvalue = (vf_coef * 2.0 / B) * (values - returns)

    def put(p, g):
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
    put(policy.log_std, dlogstd)
    for name, dout in (("pi", dmean), ("vf", dvalue.unsqueeze(-1))):
        layers, saved = nets[name], acts[name]
        dy = dout.to(cd)
        for li in reversed(range(len(layers))):
            lin = layers[li]
            gw, gb = _splitk_wgrad(dy, saved[li], split_k)
            put(lin.weight, gw)
            put(lin.bias, gb)
            if li > 0:
                dy = (dy @ lin.weight.to(cd)) * (saved[li] > 0).to(cd)
    return pl, vl


def _colsum(x):
    """fp32 column sums over the second-to-last dim as a ones-row GEMM (ATen's multi-block column reduction is
    not replay-safe inside a hipGraph on this stack; see rl/policy.py:_LinearGemmBias)."""
    xf = x.float()
    ones = torch.ones((*xf.shape[:-2], 1, xf.shape[-2]), device=x.device)
    return torch.matmul(ones, xf).squeeze(-2)


def flatten_parameters(policy, pad: int = 64):
    """Re-home every parameter (and its .grad) as a view into ONE flat fp32 vector, each slot padded to
    `pad` elements (keeps bf16 GEMM operands 128-byte aligned).  Lets the optimiser be one kernel
    (``myo_adam_clip_step``) and the data-parallel all-reduce run in place on one buffer.  Must run
    before any hipGraph captures the parameters' addresses."""
    if getattr(policy, "_flat", None) is not None:
        return policy._flat
    params = list(policy.parameters())
    # interleave actor / critic trunk layers (and the two LSTMs of a recurrent policy): adjacent slots -> stacked [2,out,in] views
    pi, vf = _layers(policy.mlp_extractor.policy_net), _layers(policy.mlp_extractor.value_net)
    lead = []
    if len(pi) == len(vf) and all(a.weight.shape == b.weight.shape for a, b in zip(pi, vf)):
        for a, b in zip(pi, vf):
            lead += [a.weight, b.weight, a.bias, b.bias]
    la, lc = getattr(policy, "lstm_actor", None), getattr(policy, "lstm_critic", None)
    if policy.recurrent and lc is not None and la.hidden_size == lc.hidden_size:
        for key in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            lead += [getattr(la, key), getattr(lc, key)]
    ids = {id(p) for p in lead}
    params = lead + [p for p in params if id(p) not in ids]
    dev = params[0].device
    slots, n = [], 0
    for p in params:
        slots.append((n, p.numel()))
        n += (p.numel() + pad - 1) // pad * pad
    pflat, gflat = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for p, (off, k) in zip(params, slots):
        pflat[off:off + k].copy_(p.data.reshape(-1))
        p.data = pflat[off:off + k].view(p.shape)
        p.grad = gflat[off:off + k].view(p.shape)
    policy._flat = {"p": pflat, "g": gflat, "slots": slots, "params": params}
    return policy._flat


class FlatAdam:
    """clip_grad_norm_ + Adam over the flat vector in libmyobatch (2 launches, capturable)."""

    def __init__(self, flat, lib, lr, max_norm, betas=(0.9, 0.999), eps=1e-5):
        self.flat, self.lib = flat, lib
        self.lr, self.max_norm, self.betas, self.eps = float(lr), float(max_norm), betas, float(eps)
        dev = flat["p"].device
        self.m, self.v = torch.zeros_like(flat["p"]), torch.zeros_like(flat["p"])
        self._step = torch.zeros(2, dtype=torch.int32, device=dev)    # [committed, taken]
        self.sq = torch.zeros(512, device=dev)   # partial sums of |g|^2 (64 of myo_adam_clip_step, or those myo_ppo_mlp_step leaves)
        self.presummed = 0        # > 0: the gradient's producer has left that many partial sums in `sq` and advanced the step counter
        self.shadow = None        # bf16 copy of the flat parameters kept in step by the Adam kernel (FusedPPOStep sets it)

    @property
    def step_count(self):
        return self._step[1]

    def set_step_count(self, step: float) -> None:
        """Adam's step counter (bias correction) when the moments are restored from a checkpoint."""
        self._step.fill_(int(step))

    def step(self, grad_scale: float = 1.0):
        f = self.flat
        stream = torch.cuda.current_stream(f["p"].device).cuda_stream
        p = lambda t: C.c_void_p(t.data_ptr())
        if self.presummed > 0:      # single-rank fused step: myo_ppo_mlp_step has done the squares and the step counter
            self.lib.check(self.lib.L.myo_adam_apply(
                p(f["p"]), p(f["g"]), p(self.m), p(self.v), f["p"].numel(), self.lr, self.betas[0], self.betas[1],
                self.eps, self.max_norm, float(grad_scale), p(self._step), p(self.sq), int(self.presummed),
                p(self.shadow) if self.shadow is not None else None, C.c_void_p(stream)))
            return
        self.lib.check(self.lib.L.myo_adam_clip_step(
            p(f["p"]), p(f["g"]), p(self.m), p(self.v), f["p"].numel(), self.lr, self.betas[0], self.betas[1],
            self.eps, self.max_norm, float(grad_scale), p(self._step), p(self.sq),
            p(self.shadow) if self.shadow is not None else None, C.c_void_p(stream)))

    def snapshot(self):
        return [t.clone() for t in (self.flat["p"], self.m, self.v, self._step)]

    def restore(self, snap):
        for t, s in zip((self.flat["p"], self.m, self.v, self._step), snap):
            t.copy_(s)
        if self.shadow is not None:
            self.shadow.copy_(self.flat["p"])


class FusedPPOStep:
    """GPU path: persistent bf16 weight copies + the HIP loss kernel.  All shapes static so the
    sequence can be captured in a hipGraph."""

    def __init__(self, policy, lib, clip_range, ent_coef, vf_coef, split_k=64):
        self.policy, self.lib = policy, lib
        self.clip, self.ent, self.vf, self.split = float(clip_range), float(ent_coef), float(vf_coef), split_k
        self.adam_syncs_shadow = False
        self.nets = _nets(policy)
        dev = policy.log_std.device
        self.master, self.half = [], []
        self.wb = {}
        flat = getattr(policy, "_flat", None)
        if flat is not None:             # one bf16 shadow of the flat vector: a single cast per step
            hflat = torch.empty_like(flat["p"], dtype=torch.bfloat16)
            self.master, self.half = [flat["p"]], [hflat]
            for p, (off, k) in zip(flat["params"], flat["slots"]):
                self.wb[id(p)] = hflat[off:off + k].view(p.shape)
        else:
            for name, layers in self.nets.items():
                for lin in layers:
                    for p in (lin.weight, lin.bias):
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                        h = torch.empty_like(p, dtype=torch.bfloat16)
                        self.master.append(p.data)
                        self.half.append(h)
                        self.wb[id(p)] = h
        if policy.log_std.grad is None:
            policy.log_std.grad = torch.zeros_like(policy.log_std)
        A = policy.act_dim
        self.merged = None
        if flat is not None:
            self.merged = self._stacked_views(flat, hflat)
        self.acc = torch.zeros(2 * A + 4, device=dev)
        self.stats = torch.zeros(2, device=dev)
        self.external_adv_stats = False      # True: the caller fills self.stats (global / no advantage normalisation)
        self.A = A
        self._work = {}
        self._mfma = {}                      # per minibatch size: (descriptor, workspace) of myo_ppo_mlp_step
        self.use_mfma_step = True            # the one-launch-per-stage path (csrc/myo_ppo_mlp.h) whenever the shapes fit
        self.adam = None                     # FlatAdam whose |g|^2 partials / step counter the fused step fills (single rank; PPO sets it)

    # ---- fused forward / loss / backward on the matrix cores (libmyobatch: myo_ppo_mlp_step)
    def _mfma_desc(self, B, obs_all, act_all, oldlp_all, adv_all, ret_all, idx):
        """Descriptor of myo_ppo_mlp_step for minibatch size B, or None when the architecture / batch size has no fused path
        (then the GEMM-per-layer path below runs).  The descriptor and its workspace are cached: hipGraph-safe."""
        from .. import native
        flat = getattr(self.policy, "_flat", None)
        if not self.use_mfma_step or flat is None or self.merged is None or len(self.merged) != 2 or self.policy.recurrent:
            return None
        pi, vf = self.nets["pi"], self.nets["vf"]
        O = obs_all.shape[1]
        hid = pi[0].weight.shape[0]
        shapes_ok = all(tuple(l.weight.shape) == s for l, s in ((pi[0], (hid, O)), (vf[0], (hid, O)), (pi[1], (hid, hid)), (vf[1], (hid, hid)),
                                                               (pi[2], (self.A, hid)), (vf[2], (1, hid))))
        if not shapes_ok or self.policy.log_std.dim() != 1 or obs_all.dtype != torch.float32:
            return None
        G = flat["p"].numel()
        # one descriptor + workspace per SHAPE (the workspace is ~90 MB at B = 16384); the six data pointers are rewritten on every
        # call — the library reads the descriptor at call time, so a captured graph keeps the pointers it was captured with and an
        # eager caller may pass fresh tensors per minibatch without growing the cache
        key = (B, O, self.A, hid, G)
        hit = self._mfma.get(key)
        if hit is not None:
            d = hit[0]
            if d is not None:
                d.obs, d.act, d.oldlp, d.adv, d.ret, d.idx = (t.data_ptr() for t in (obs_all, act_all, oldlp_all, adv_all, ret_all, idx))
            return d
        nbytes = self.lib.L.myo_ppo_mlp_workspace_bytes(B, O, self.A, hid, G)
        if nbytes <= 0:
            self._mfma[key] = (None, None)
            return None
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=obs_all.device)
        slot = {id(p): sl[0] for p, sl in zip(flat["params"], flat["slots"])}
        d = native.PpoMlpDesc()
        d.obs, d.act, d.oldlp, d.adv, d.ret, d.idx = (t.data_ptr() for t in (obs_all, act_all, oldlp_all, adv_all, ret_all, idx))
        d.B, d.O, d.A, d.hidden = B, O, self.A, hid
        d.params, d.grads, d.G = flat["p"].data_ptr(), flat["g"].data_ptr(), G
        for k, (a, b) in enumerate(zip(pi, vf)):
            for net, lin in enumerate((a, b)):
                wname, bname = (("off_W1", "off_b1"), ("off_W2", "off_b2"), ("off_Wh", "off_bh"))[k]
                getattr(d, wname)[net] = slot[id(lin.weight)]
                getattr(d, bname)[net] = slot[id(lin.bias)]
        d.off_log_std = slot[id(self.policy.log_std)]
        d.clip, d.vf_coef, d.ent_coef = self.clip, self.vf, self.ent
        d.adv_stats, d.acc = self.stats.data_ptr(), self.acc.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
        if self.adam is not None and self.lib.L.myo_ppo_mlp_sqnorm_parts(self.A) <= self.adam.sq.numel():
            d.sqnorm_part, d.adam_step = self.adam.sq.data_ptr(), self.adam._step.data_ptr()
        self._mfma[key] = (d, ws)
        return d

    # ---- rollout inference + sampling on the matrix cores (libmyobatch: myo_ppo_mlp_rollout)
    def rollout_desc(self, obs, seed, draw, t_idx, obs_buf, act_buf, val_buf, logp_buf, clipped):
        """Descriptor of myo_ppo_mlp_rollout for the policy input `obs` [N,O] (static tensors of the rollout graph), or None
        when the architecture / number of envs has no fused path.  Cached; `refresh_rollout_images` rebuilds the weight
        images it reads and has to run after every parameter change."""
        from .. import native
        flat = getattr(self.policy, "_flat", None)
        if not self.use_mfma_step or flat is None or self.merged is None or len(self.merged) != 2 or self.policy.recurrent:
            return None
        pi, vf = self.nets["pi"], self.nets["vf"]
        N, O = obs.shape
        hid = pi[0].weight.shape[0]
        shapes_ok = all(tuple(l.weight.shape) == s for l, s in ((pi[0], (hid, O)), (vf[0], (hid, O)), (pi[1], (hid, hid)), (vf[1], (hid, hid)),
                                                               (pi[2], (self.A, hid)), (vf[2], (1, hid))))
        if not shapes_ok or self.policy.log_std.dim() != 1 or obs.dtype != torch.float32 or N % 32:
            return None
        nbytes = self.lib.L.myo_ppo_mlp_rollout_workspace_bytes(O, self.A, hid)
        if nbytes <= 0:
            return None
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=obs.device)
        slot = {id(p): sl[0] for p, sl in zip(flat["params"], flat["slots"])}
        d = native.PpoMlpRolloutDesc()
        d.obs, d.N, d.O, d.A, d.hidden, d.params = obs.data_ptr(), N, O, self.A, hid, flat["p"].data_ptr()
        for k, (a, b) in enumerate(zip(pi, vf)):
            for net, lin in enumerate((a, b)):
                wname, bname = (("off_W1", "off_b1"), ("off_W2", "off_b2"), ("off_Wh", "off_bh"))[k]
                getattr(d, wname)[net] = slot[id(lin.weight)]
                getattr(d, bname)[net] = slot[id(lin.bias)]
        d.off_log_std = slot[id(self.policy.log_std)]
        d.seed, d.draw_counter, d.t_idx = seed, draw.data_ptr(), t_idx.data_ptr()
        d.obs_buf, d.act_buf, d.val_buf, d.logp_buf, d.clipped = (t.data_ptr() for t in (obs_buf, act_buf, val_buf, logp_buf, clipped))
        d.deterministic, d.workspace, d.workspace_bytes = 0, ws.data_ptr(), nbytes
        self._rollout = (d, ws)
        self.refresh_rollout_images()
        return d

    def refresh_rollout_images(self):
        """bf16 weight images of the rollout kernel from the current master weights (one launch; call after every update)."""
        r = getattr(self, "_rollout", None)
        if r is not None:
            stream = torch.cuda.current_stream(self.acc.device).cuda_stream
            self.lib.check(self.lib.L.myo_ppo_mlp_rollout_refresh(C.byref(r[0]), C.c_void_p(stream)))

    def rollout_policy(self, d):
        stream = torch.cuda.current_stream(self.acc.device).cuda_stream
        self.lib.check(self.lib.L.myo_ppo_mlp_rollout(C.byref(d), C.c_void_p(stream)))

    def _mfma_step(self, d):
        d.compute_adv_stats = 0 if self.external_adv_stats else 1
        stream = torch.cuda.current_stream(self.acc.device).cuda_stream
        self.lib.check(self.lib.L.myo_ppo_mlp_step(C.byref(d), C.c_void_p(stream)))
        if self.adam is not None:           # the Adam call that follows takes the partial sums this step left (or does its own)
            self.adam.presummed = self.lib.L.myo_ppo_mlp_sqnorm_parts(self.A) if d.sqnorm_part else 0
        return self.acc[self.A], self.acc[self.A + 1]

    def _workbuf(self, key, n):
        """Zero-initialised scratch of the block-ticket kernels (allocated once per shape: graph-safe)."""
        w = self._work.get((key, n))
        if w is None:
            w = self._work[(key, n)] = torch.zeros(n, device=self.acc.device)
        return w

    def _stacked_views(self, flat, hflat):
        """[2,out,in] / [2,out] views over the adjacent actor/critic trunk slots (None if not adjacent)."""
        pi, vf = self.nets["pi"][:-1], self.nets["vf"][:-1]
        if len(pi) != len(vf) or (not pi and not self.policy.recurrent):
            return None                  # (a recurrent policy may have no trunk at all: the heads sit on the LSTM outputs)
        slot = {id(p): sl for p, sl in zip(flat["params"], flat["slots"])}
        out = []
        for a, b in zip(pi, vf):
            lay = {}
            for key, pa, pb in (("w", a.weight, b.weight), ("b", a.bias, b.bias)):
                (oa, k), (ob, kb) = slot[id(pa)], slot[id(pb)]
                if pa.shape != pb.shape or ob != oa + k:
                    return None
                shape = (2,) + tuple(pa.shape)
                lay[key + "h"] = hflat[oa:oa + 2 * k].view(shape)
                lay[key + "g"] = flat["g"][oa:oa + 2 * k].view(shape)
            out.append(lay)
        return out

    def _reduce(self, part, out, groups, splits):
        """out[g] = sum_k part[g*splits + k] (bf16 or fp32 partials -> fp32) with the HIP split-K reducer."""
        n = out.numel() // groups
        stream = torch.cuda.current_stream(out.device).cuda_stream
        self.lib.check(self.lib.L.myo_splitk_reduce(C.c_void_p(part.data_ptr()), int(part.dtype == torch.bfloat16),
                                                    C.c_void_p(out.data_ptr()), groups, splits, n, C.c_void_p(stream)))

    def _reduce2(self, part_a, out_a, groups_a, splits_a, part_b, out_b, groups_b, splits_b):
        """Two _reduce calls in one launch."""
        stream = torch.cuda.current_stream(out_a.device).cuda_stream
        self.lib.check(self.lib.L.myo_splitk_reduce2(
            C.c_void_p(part_a.data_ptr()), int(part_a.dtype == torch.bfloat16), C.c_void_p(out_a.data_ptr()), groups_a, splits_a,
            out_a.numel() // groups_a, C.c_void_p(part_b.data_ptr()), int(part_b.dtype == torch.bfloat16),
            C.c_void_p(out_b.data_ptr()), groups_b, splits_b, out_b.numel() // groups_b, C.c_void_p(stream)))

    def _relu_bwd_partial(self, dh, act):
        """dh *= (act > 0) in place and the per-32-row column sums of the result (fp32 [G*B/32, H]); None if the
        shape has no fast path (the caller then uses _relu_bwd_bias)."""
        G, B, H = dh.shape
        if not (B % 32 == 0 and H % 16 == 0 and 256 % (H // 2) == 0):
            return None
        partial = torch.empty((G * B // 32, H), device=dh.device)
        stream = torch.cuda.current_stream(dh.device).cuda_stream
        self.lib.check(self.lib.L.myo_relu_bwd_colsum_bf16(C.c_void_p(dh.data_ptr()), C.c_void_p(act.data_ptr()), G * B, H,
                                                           C.c_void_p(partial.data_ptr()), C.c_void_p(stream)))
        return partial

    def _relu_bwd_bias(self, dh, act, bias_grad):
        """dh *= (act > 0) in place; bias_grad[g] = column sums of dh[g]  (dh, act: bf16 [G, B, H])."""
        G, B, H = dh.shape
        if B % 32 == 0 and H % 16 == 0 and 256 % (H // 2) == 0:
            partial = torch.empty((G * B // 32, H), device=dh.device)
            stream = torch.cuda.current_stream(dh.device).cuda_stream
            self.lib.check(self.lib.L.myo_relu_bwd_colsum_bf16(C.c_void_p(dh.data_ptr()), C.c_void_p(act.data_ptr()), G * B, H,
                                                               C.c_void_p(partial.data_ptr()), C.c_void_p(stream)))
            self._reduce(partial, bias_grad, G, B // 32)
            return dh
        dh = torch.ops.aten.threshold_backward(dh, act, 0)
        bias_grad.copy_(_colsum(dh))
        return dh

    def _loss_kernel(self, mean, values, actions, old_logp, adv, returns, dmean_h=None, dvalue_h=None, direct=None):
        """direct = (g_log_std, g_bias_pi, g_bias_vf) tensors written by the finish kernel (merged path)."""
        pol = self.policy
        B, A = mean.shape[0], self.A
        in_bf16 = int(mean.dtype == torch.bfloat16)
        assert values.dtype == mean.dtype
        d = direct or (None, None, None)
        dev = mean.device
        dmean, dvalue = torch.empty((B, A), device=dev), torch.empty(B, device=dev)
        work = self._workbuf("loss", ((B + 63) // 64) * (2 * A + 3))
        stream = torch.cuda.current_stream(dev).cuda_stream
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self.lib.check(self.lib.L.myo_ppo_loss_grad(
            p(mean), p(values), p(actions), p(old_logp), p(adv), p(returns), p(pol.log_std.data), p(self.stats),
            B, A, self.clip, self.vf, p(dmean), p(dvalue), p(self.acc), p(dmean_h), p(dvalue_h), p(work), in_bf16, self.ent,
            p(d[0]), p(d[1]), p(d[2]), C.c_void_p(stream)))
        return dmean, dvalue

    def _sde_loss(self, mean_h, value_h, lat_h, actions, old_logp, adv, returns):
        """The PPO loss and its gradients for generalised state-dependent exploration (SB3 StateDependentNoiseDistribution with
        learn_features=False: /root/reference/docs/summary.md:100; rl/policy.py evaluate_actions): the action variance of sample i is
        var_ia = sum_l lat_il^2 exp(2 log_std_la) + eps with the latent DETACHED, so the trunk sees the loss through the mean only and
        log_std [latent, act] through the variance (log-probability and per-sample entropy).  fp32 ATen ops on the bf16 trunk outputs —
        capturable (every sum over the batch is a ones-row GEMM, see _colsum); writes log_std.grad and the head-bias gradients, leaves
        (policy loss, value loss) in acc[A], acc[A+1], returns bf16 d(loss)/d(mean) [B,A] and d(loss)/d(value) [B,1]."""
        pol, A, B = self.policy, self.A, mean_h.shape[0]
        pi_head, vf_head = self.nets["pi"][-1], self.nets["vf"][-1]
        lat2 = lat_h.float() ** 2                                     # [B, L]
        e2 = torch.exp(2.0 * pol.log_std.data.float())                # [L, A]
        var = lat2 @ e2 + pol.SDE_EPS
        diff = actions - mean_h.float()
        z2 = diff * diff
        logp = (-0.5 * z2 / var - 0.5 * torch.log(var)).sum(-1) - 0.5 * A * math.log(2 * math.pi)
        advn = (adv - self.stats[0]) / (self.stats[1] + 1e-8)
        ratio = torch.exp(logp - old_logp)
        s1, s2 = advn * ratio, advn * torch.clamp(ratio, 1 - self.clip, 1 + self.clip)
        inside = (ratio > 1 - self.clip) & (ratio < 1 + self.clip)
        dlogp = -(advn * ratio) * torch.where(s1 <= s2, torch.ones_like(ratio), inside.to(ratio.dtype)) / B
        dmean = dlogp.unsqueeze(-1) * diff / var
        dvar = dlogp.unsqueeze(-1) * (0.5 * z2 / (var * var) - 0.5 / var) - (self.ent / B) * 0.5 / var      # entropy: sum_a 0.5 log var + const
        pol.log_std.grad.copy_(2.0 * e2 * (lat2.t() @ dvar))
        v = value_h.float().reshape(B)
        err = v - returns
        dvalue = (self.vf * 2.0 / B) * err
        sums = _colsum(torch.stack([torch.min(s1, s2), err * err], -1))          # [2]: sum of the surrogate, sum of squared errors
        self.acc[A].copy_(-sums[0] / B)
        self.acc[A + 1].copy_(sums[1] / B)
        pi_head.bias.grad.copy_(_colsum(dmean))
        vf_head.bias.grad.copy_(_colsum(dvalue.unsqueeze(-1)))
        return dmean.to(torch.bfloat16), dvalue.to(torch.bfloat16).unsqueeze(-1)

    @torch.no_grad()
    def run_indexed(self, obs_all, act_all, oldlp_all, adv_all, ret_all, idx):
        """One minibatch step on rows `idx` of the rollout arrays (merged path: HIP gather kernel)."""
        if self.merged is None:
            return self.run(obs_all[idx], act_all[idx], oldlp_all[idx], adv_all[idx], ret_all[idx])
        d = self._mfma_desc(idx.shape[0], obs_all, act_all, oldlp_all, adv_all, ret_all, idx)
        if d is not None:
            return self._mfma_step(d)
        if self.adam is not None:
            self.adam.presummed = 0          # GEMM path below: Adam squares the gradient itself
        B, O, A = idx.shape[0], obs_all.shape[1], self.A
        dev = obs_all.device
        x2 = torch.empty((2, B, O), device=dev, dtype=torch.bfloat16)
        act, oldlp = torch.empty((B, A), device=dev), torch.empty(B, device=dev)
        adv, ret = torch.empty(B, device=dev), torch.empty(B, device=dev)
        work = self._workbuf("gather", 2 * ((B + 15) // 16))
        stream = torch.cuda.current_stream(dev).cuda_stream
        p = lambda t: C.c_void_p(t.data_ptr())
        self.lib.check(self.lib.L.myo_ppo_gather(p(obs_all), p(act_all), p(oldlp_all), p(adv_all), p(ret_all), p(idx), B, O, A,
                                                 p(x2), 2, p(act), p(oldlp), p(adv), p(ret), None if self.external_adv_stats else p(self.stats), p(work),
                                                 C.c_void_p(stream)))
        return self._merged_core(x2, act, oldlp, adv, ret)

    @torch.no_grad()
    def _run_merged(self, obs, actions, old_logp, adv, returns):
        if self.use_mfma_step:
            B = obs.shape[0]
            ar = self._work.get(("arange", B))
            if ar is None:
                ar = self._work[("arange", B)] = torch.arange(B, device=obs.device)
            d = self._mfma_desc(B, obs.contiguous(), actions.contiguous(), old_logp.contiguous(), adv.contiguous(), returns.contiguous(), ar)
            if d is not None:
                return self._mfma_step(d)
        if self.adam is not None:
            self.adam.presummed = 0
        var, mu = torch.var_mean(adv)
        self.stats[0].copy_(mu)
        self.stats[1].copy_(var.sqrt())
        x2 = obs.to(torch.bfloat16).unsqueeze(0).expand(2, obs.shape[0], obs.shape[1]).contiguous()
        return self._merged_core(x2, actions, old_logp, adv, returns)

    def refresh_shadow(self):
        """bf16 shadow of the master weights (one cast kernel on the flat vector)."""
        if len(self.half) == 1:
            self.half[0].copy_(self.master[0])       # flat vector: one cast kernel
        else:
            torch._foreach_copy_(self.half, self.master)
        self.refresh_rollout_images()

    def trunk_heads(self, x2):
        """Stacked actor/critic forward on the bf16 shadow weights: x2 bf16 [2,B,obs] -> (hidden
        activations per layer, action mean bf16 [B,A], value bf16 [B,1])."""
        B = x2.shape[1]
        stream = torch.cuda.current_stream(x2.device).cuda_stream
        h, saved = x2, [x2]
        for lay in self.merged:
            h = torch.bmm(h, lay["wh"].transpose(1, 2))
            self.lib.check(self.lib.L.myo_bias_relu_bf16(C.c_void_p(h.data_ptr()), C.c_void_p(lay["bh"].data_ptr()), 2, B,
                                                         h.shape[2], C.c_void_p(stream)))
            saved.append(h)
        pi_head, vf_head = self.nets["pi"][-1], self.nets["vf"][-1]
        mean_h = torch.addmm(self.wb[id(pi_head.bias)], h[0], self.wb[id(pi_head.weight)].t())
        value_h = torch.addmm(self.wb[id(vf_head.bias)], h[1], self.wb[id(vf_head.weight)].t())
        return saved, mean_h, value_h

    def _merged_core(self, x2, actions, old_logp, adv, returns, want_dx=False):
        """Actor and critic trunks as ONE batched GEMM per layer (batch = net), heads separate.
        x2: bf16 [2, B, in] (the same minibatch twice, or the two LSTM outputs of a recurrent policy); self.stats holds
        the advantage moments.  want_dx: also return d(loss)/d(x2) (bf16 [2, B, in])"""
        pol, L = self.policy, self.merged
        B, A = x2.shape[1], self.A
        s = self.split if (B % self.split == 0) else 1
        dev = x2.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        if not self.adam_syncs_shadow:       # (PPO: the Adam kernel rewrites the bf16 shadow with the parameters)
            self.refresh_shadow()
        saved, mean_h, value_h = self.trunk_heads(x2)
        h = saved[-1]
        pi_head, vf_head = self.nets["pi"][-1], self.nets["vf"][-1]
        wpi, wvf = self.wb[id(pi_head.weight)], self.wb[id(vf_head.weight)]
        acc = self.acc
        if getattr(pol, "use_sde", False):
            dmean_h, dvalue_h = self._sde_loss(mean_h, value_h, h[0], actions, old_logp, adv, returns)
        else:
            dmean_h = torch.empty((B, A), device=dev, dtype=torch.bfloat16)
            dvalue_h = torch.empty((B, 1), device=dev, dtype=torch.bfloat16)
            self._loss_kernel(mean_h, value_h, actions, old_logp, adv, returns, dmean_h, dvalue_h,
                              direct=(pol.log_std.grad, pi_head.bias.grad, vf_head.bias.grad))
        # heads: dW (split-K) and the gradient entering the trunks
        parts = [torch.bmm(dy.view(s, B // s, -1).transpose(1, 2), x.view(s, B // s, -1))
                 for dy, x in ((dmean_h, h[0]), (dvalue_h, h[1]))]
        if pi_head.weight.grad.numel() % 2 == 0 and vf_head.weight.grad.numel() % 2 == 0:
            self._reduce2(parts[0], pi_head.weight.grad, 1, s, parts[1], vf_head.weight.grad, 1, s)
        else:
            self._reduce(parts[0], pi_head.weight.grad, 1, s)
            self._reduce(parts[1], vf_head.weight.grad, 1, s)
        dh = torch.empty_like(h)
        torch.mm(dmean_h, wpi, out=dh[0])
        torch.mm(dvalue_h, wvf, out=dh[1])
        for li in reversed(range(len(L))):
            lay, x = L[li], saved[li]
            bias_part = self._relu_bwd_partial(dh, saved[li + 1])
            if bias_part is None:
                dh = self._relu_bwd_bias(dh, saved[li + 1], lay["bg"])
            H, I = lay["wh"].shape[1], lay["wh"].shape[2]
            part = torch.bmm(dh.view(2 * s, B // s, H).transpose(1, 2), x.reshape(2 * s, B // s, I))
            if bias_part is not None and (H * I) % 2 == 0:      # weight partials and bias partials: one launch
                self._reduce2(part, lay["wg"], 2, s, bias_part, lay["bg"], 2, B // 32)
            else:
                if bias_part is not None:
                    self._reduce(bias_part, lay["bg"], 2, B // 32)
                self._reduce(part, lay["wg"], 2, s)
            if li > 0 or want_dx:
                dh = torch.bmm(dh, lay["wh"])
        if want_dx:
            return acc[A], acc[A + 1], dh
        return acc[A], acc[A + 1]

    @torch.no_grad()
    def run(self, obs, actions, old_logp, adv, returns):
        if self.merged is not None:
            return self._run_merged(obs, actions, old_logp, adv, returns)
        pol = self.policy
        if self.adam is not None:
            self.adam.presummed = 0
        B, A = obs.shape[0], self.A
        torch._foreach_copy_(self.half, self.master)
        x0 = obs.to(torch.bfloat16)
        acts, outs = {}, {}
        for name, layers in self.nets.items():
            h = x0
            saved = [h]
            for li, lin in enumerate(layers):
                h = torch.addmm(self.wb[id(lin.bias)], h, self.wb[id(lin.weight)].t())
                if li < len(layers) - 1:
                    h = torch.relu_(h)
                    saved.append(h)
            acts[name], outs[name] = saved, h.float()
        mean, values = outs["pi"].contiguous(), outs["vf"].reshape(B).contiguous()
        var, mu = torch.var_mean(adv)                      # unbiased, as torch.std in SB3
        self.stats[0].copy_(mu)
        self.stats[1].copy_(var.sqrt())
        dmean, dvalue = self._loss_kernel(mean, values, actions, old_logp, adv, returns)
        torch.sub(self.acc[:A], self.ent, out=pol.log_std.grad)
        for name, dout in (("pi", dmean), ("vf", dvalue.unsqueeze(-1))):
            layers, saved = self.nets[name], acts[name]
            dy = dout.to(torch.bfloat16)
            for li in reversed(range(len(layers))):
                lin = layers[li]
                x = saved[li]
                s = self.split if (B % self.split == 0) else 1
                part = torch.bmm(dy.view(s, B // s, -1).transpose(1, 2), x.view(s, B // s, -1))
                torch.sum(part, 0, dtype=torch.float32, out=lin.weight.grad)
                lin.bias.grad.copy_(_colsum(dy))
                if li > 0:
                    dy = torch.ops.aten.threshold_backward(dy @ self.wb[id(lin.weight)], saved[li], 0)
        return self.acc[A], self.acc[A + 1]
