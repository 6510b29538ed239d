import sys, time, torch
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.policy import ActorCriticPolicy
from myochallenge_amd.rl.ppo import PPO, PPOConfig
from myochallenge_amd.rl.vec_normalize import VecNormalize
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = EnvironmentFactory.create("CustomMyoReorientP1", num_envs=N, seed=1)
torch.manual_seed(0)
pol = ActorCriticPolicy(env.obs_dim, env.act_dim, (256, 256), (256, 256), lstm_hidden_size=256, log_std_init=-2.0)
algo = PPO(VecNormalize(env), pol, PPOConfig(n_steps=32, batch_size=N * 32 // 8, n_epochs=4, learning_rate=2.5e-5))
fin = lambda: {n: bool(torch.isfinite(p).all()) for n, p in pol.named_parameters() if not torch.isfinite(p).all()}
for it in range(4):
    t0 = time.time(); algo.collect_rollouts(); torch.cuda.synchronize(); print("collect s", time.time() - t0)
    print("it", it, "bufs finite", [bool(torch.isfinite(getattr(algo, b)).all()) for b in ("obs_buf", "act_buf", "rew_buf", "val_buf", "logp_buf")],
          "done frac", float(env._done.float().mean()), "state0 finite", [bool(torch.isfinite(x).all()) for x in algo._rollout_state0])
    from myochallenge_amd.rl.ppo import compute_gae
    adv, ret = compute_gae(algo.rew_buf, algo.val_buf, algo.start_buf, algo._last_values, algo._last_starts, 0.99, 0.95)
    print("   last_values finite", bool(torch.isfinite(algo._last_values).all()), "adv/ret finite", bool(torch.isfinite(adv).all()), bool(torch.isfinite(ret).all()),
          "starts mid-rollout", int(algo.start_buf[1:].sum()))
    st = algo.train(); torch.cuda.synchronize()
    print("   train", st, "nonfinite params", fin(), "grad finite", bool(torch.isfinite(algo._flat_grad).all()), "m,v finite",
          bool(torch.isfinite(algo._flat_adam.m).all()), bool(torch.isfinite(algo._flat_adam.v).all()))
