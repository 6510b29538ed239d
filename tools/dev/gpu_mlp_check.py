"""Fused MFMA PPO step (myo_ppo_mlp_step) vs the GEMM-per-layer path and the fp32 reference, plus timings."""
import copy, time, torch
from myochallenge_amd import native
from myochallenge_amd.rl.fused_mlp import FusedPPOStep, flatten_parameters, ppo_mlp_step_grads
from myochallenge_amd.rl.policy import ActorCriticPolicy
lib = native.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B in (4096, 16384):
    pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=None).to(dev)
    ref_pol = copy.deepcopy(pol)
    obs = torch.randn(B, 86, device=dev)
    with torch.no_grad():
        act = pol.act(obs, None, None)[0]
        oldlp = pol.evaluate_actions(obs, act)[1] + torch.randn(B, device=dev) * 0.05
    adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)
    pl_ref, vl_ref = ppo_mlp_step_grads(ref_pol, obs, act, oldlp, adv, ret, 0.2, 0.01, 0.7, bf16=False)
    ref = [p.grad.clone() for p in ref_pol.parameters()]
    flatten_parameters(pol)
    step = FusedPPOStep(pol, lib, 0.2, 0.01, 0.7)
    res = {}
    for name, flag in (("gemm", False), ("mfma", True)):
        step.use_mfma_step = flag
        pol._flat["g"].zero_()
        pl, vl = step.run(obs, act, oldlp, adv, ret)
        torch.cuda.synchronize()
        res[name] = ([p.grad.clone() for p in pol.parameters()], float(pl), float(vl))
        errs = {n: float((g - r).norm() / (r.norm() + 1e-12)) for (n, _), g, r in zip(pol.named_parameters(), res[name][0], ref)}
        print(B, name, "pl %.6f (ref %.6f) vl %.6f (ref %.6f)" % (float(pl), float(pl_ref), float(vl), float(vl_ref)))
        print("   rel err vs fp32 ref:", {k: round(v, 4) for k, v in errs.items()})
    d = {n: float((a - b).norm() / (b.norm() + 1e-12)) for (n, _), a, b in zip(pol.named_parameters(), res["mfma"][0], res["gemm"][0])}
    print("   mfma vs gemm path:", {k: round(v, 4) for k, v in d.items()})
    # determinism of the fused path
    step.use_mfma_step = True
    step.run(obs, act, oldlp, adv, ret); torch.cuda.synchronize(); g1 = pol._flat["g"].clone()
    step.run(obs, act, oldlp, adv, ret); torch.cuda.synchronize(); g2 = pol._flat["g"].clone()
    print("   bitwise repeatable:", torch.equal(g1, g2))
    for name, flag in (("gemm", False), ("mfma", True)):
        step.use_mfma_step = flag
        for _ in range(3): step.run(obs, act, oldlp, adv, ret)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(50): step.run(obs, act, oldlp, adv, ret)
        torch.cuda.synchronize(); print("   %s eager ms/step %.3f" % (name, (time.time() - t) / 50 * 1e3))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): step.run(obs, act, oldlp, adv, ret)
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(200): g.replay()
        torch.cuda.synchronize(); print("   %s graph ms/step %.3f" % (name, (time.time() - t) / 200 * 1e3))
