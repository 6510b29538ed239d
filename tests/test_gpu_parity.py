"""-m gpu: the HIP kernels on a real MI355X against the oracle, through the C ABI.
Same cases as test_emu_parity.py plus full-size (4096 env) property checks."""
import numpy as np
import pytest

import parity_cases as pc
from myochallenge_amd import native

pytestmark = pytest.mark.gpu


def test_native_library_is_the_hip_build(hip_lib):
    assert "gfx950" in hip_lib.version and not hip_lib.is_emulation


def test_forward_stages_f64(hip_lib, models):
    pc.case_forward_stages(hip_lib, models, native.MYO_F64, 1e-9)


def test_forward_stages_f32(hip_lib, models):
    pc.case_forward_stages(hip_lib, models, native.MYO_F32, 2e-4)      # north_star: 1e-4 rel


@pytest.mark.parametrize("name,integ,steps", [("finger", 1, 120), ("load", None, 300), ("finger", None, 60)])
def test_trajectory_small_models(hip_lib, models, name, integ, steps):
    pc.case_trajectory(hip_lib, models[name], steps, native.MYO_F64, 1e-8, integrator=integ)


@pytest.mark.parametrize("integ,steps", [(None, 60), (1, 25)])
def test_trajectory_hand(hip_lib, models, integ, steps):
    q = models["hand"].qpos0.copy(); q[0] = -1.57
    pc.case_trajectory(hip_lib, models["hand"], steps, native.MYO_F64, 1e-8, integrator=integ, q0=q)


def test_task_step_f64(hip_lib, models):
    pc.case_task_step(hip_lib, models, native.MYO_F64, 1e-7)


def test_task_step_f32_single_steps(hip_lib, models):
    pc.case_task_step(hip_lib, models, native.MYO_F32, 1e-4, nsteps=3)


def test_vecenv_protocol(hip_lib, models):
    pc.case_vecenv_protocol(hip_lib, models, native.MYO_F64)
    pc.case_vecenv_protocol(hip_lib, models, native.MYO_F32)


def test_reset_logic(hip_lib, models):
    pc.case_reset_logic(hip_lib, models, native.MYO_F32)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_full_size_properties(hip_lib, dtype):
    """BASELINE config B size (4096 envs): size-independent invariants over a rollout with
    auto-resets: finite obs, activations in [0, sigmoid32(2.5)], err = target - object,
    unit quaternions, balls never interpenetrate by more than the soft-contact depth,
    bit-identical replay from the same seed."""
    import torch
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory

    def rollout():
        env = EnvironmentFactory.create("CustomMyoBaodingBallsP2", num_envs=4096, seed=42, dtype=dtype)
        g = torch.Generator(device="cuda"); g.manual_seed(0)
        obs = env.reset_tensor().clone()
        ndone, hist = 0, []
        for t in range(30):
            a = torch.clamp(torch.randn((4096, 39), device="cuda", generator=g) * 0.5, -1, 1)
            obs, rew, done, trunc, term, comps, ep = env.step_tensor(a)
            ndone += int(done.sum())
            assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
            assert float((obs[:, 41:44] - (obs[:, 35:38] - obs[:, 23:26])).abs().max()) < 1e-6
            act = obs[:, 47:]
            assert float(act.min()) >= 0 and float(act.max()) <= 0.9241419
            hist.append(obs.clone())
        qp, qv, ac, tm = env.get_state()
        for o in (26, 33):
            assert float((qp[:, o:o + 4].norm(dim=1) - 1).abs().max()) < 1e-6
        d = (qp[:, 23:26] - qp[:, 30:33]).norm(dim=1)
        alive = (qp[:, 25] > 1.25) & (qp[:, 32] > 1.25)
        assert float(d[alive].min()) > 0.018       # no tunnelling: 2 r_min = 0.036 minus soft-contact depth under squeeze
        env.close()
        return torch.stack(hist), ndone

    h1, n1 = rollout()
    h2, n2 = rollout()
    assert n1 > 0 and n1 == n2 and torch.equal(h1, h2)


def test_fused_ppo_step_matches_autograd(hip_lib):
    """FusedPPOStep (bf16 GEMMs + myo_ppo_loss_grad HIP kernel) vs autograd of PPO._loss in fp32."""
    import torch
    from myochallenge_amd.rl.fused_mlp import FusedPPOStep
    from myochallenge_amd.rl.policy import ActorCriticPolicy
    from myochallenge_amd.rl.ppo import PPO, PPOConfig
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=None).to(dev)
    B = 4096
    obs, act = torch.randn(B, 86, device=dev), torch.randn(B, 39, device=dev) * 0.3
    with torch.no_grad():
        oldlp = pol.evaluate_actions(obs, act)[1] + torch.randn(B, device=dev) * 0.3
    adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)

    class E:
        num_envs, obs_dim, act_dim, device = 4, 86, 39, dev
    algo = PPO(E(), pol, PPOConfig(n_steps=2, clip_range=0.2, ent_coef=0.01, vf_coef=0.7, bf16=True, use_graphs=False))
    with algo._autocast():                      # same bf16 GEMM precision as the fused path
        v, lp, ent = pol.evaluate_actions(obs, act)
    loss, pl_ref, vl_ref = algo._loss(v, lp, ent, oldlp, adv, ret)
    pol.zero_grad(); loss.backward()
    ref = [p.grad.clone() for p in pol.parameters()]
    for p in pol.parameters():
        p.grad = None
    step = FusedPPOStep(pol, hip_lib, 0.2, 0.01, 0.7)
    pl, vl = step.run(obs, act, oldlp, adv, ret)
    torch.cuda.synchronize()
    assert abs(float(pl - pl_ref)) < 2e-2 * max(1.0, abs(float(pl_ref))) and abs(float(vl - vl_ref)) < 2e-2 * float(vl_ref)
    for (name, p), r in zip(pol.named_parameters(), ref):
        err = float((p.grad - r).norm() / (r.norm() + 1e-12))
        assert err < 4e-2, (name, err)          # both sides bf16 GEMMs; different rounding points


def test_gae_kernel_matches_torch_scan(hip_lib):
    import torch
    from myochallenge_amd.rl.ppo import compute_gae
    torch.manual_seed(1)
    T, N = 17, 1000
    r, v = torch.randn(T, N), torch.randn(T, N)
    st = (torch.rand(T, N) < 0.2).float()
    lv, ld = torch.randn(N), (torch.rand(N) < 0.3).float()
    a_ref, ret_ref = compute_gae(r, v, st, lv, ld, 0.99, 0.9)                     # CPU torch loop
    a, ret = compute_gae(*(x.cuda() for x in (r, v, st, lv, ld)), 0.99, 0.9)      # myo_gae kernel
    assert float((a.cpu() - a_ref).abs().max()) < 1e-4 and float((ret.cpu() - ret_ref).abs().max()) < 1e-4


def test_flat_adam_matches_torch_clip_and_adam(hip_lib):
    """myo_adam_clip_step == clip_grad_norm_(0.5) + torch.optim.Adam(eps=1e-5) over several steps."""
    import torch
    from myochallenge_amd.rl.fused_mlp import FlatAdam, flatten_parameters
    from myochallenge_amd.rl.policy import ActorCriticPolicy
    torch.manual_seed(3)
    pol = ActorCriticPolicy(86, 39).cuda()
    ref = ActorCriticPolicy(86, 39).cuda()
    ref.load_state_dict(pol.state_dict())
    flat = flatten_parameters(pol)
    opt = FlatAdam(flat, hip_lib, 2.5e-4, 0.5)
    topt = torch.optim.Adam(ref.parameters(), lr=2.5e-4, eps=1e-5)
    for it in range(5):
        scale = 10.0 if it % 2 == 0 else 0.01          # exercises both the clipped and the unclipped branch
        for p, q in zip(pol.parameters(), ref.parameters()):
            g = torch.randn_like(p) * scale
            p.grad.copy_(g)
            q.grad = g.clone()
        opt.step(1.0)
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        topt.step()
    torch.cuda.synchronize()
    assert int(opt.step_count) == 5
    for (n, p), q in zip(pol.named_parameters(), ref.parameters()):
        assert p.data_ptr() >= flat["p"].data_ptr()
        err = float((p - q).abs().max())
        assert err < 2e-6, (n, err)
