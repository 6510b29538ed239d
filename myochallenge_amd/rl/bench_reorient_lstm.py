"""BASELINE config E (die-reorient, 4096 envs on one MI355X, recurrent LSTM policy): env-steps/s of rollout + PPO update.
Not the headline bench; `bench.py` records it as `variants.config_E_lstm256`, `tools/bench_reorient.py` prints it.  The env step is
the step kernel's MYO_TASK_REORIENT task (csrc/myo_task.h); the LSTM policy (/root/reference/src/main_reorient.py:53-71: LSTM-256
actor + critic -> [256, 256] ReLU, log_std_init -2) rolls out and trains on the HIP-kernel recurrent path."""
import time


def run(envs: int = 4096, n_steps: int = 32, iters: int = 3, env_only_steps: int = 50) -> dict:
    import torch
    from ..envs.environment_factory import EnvironmentFactory
    from .policy import ActorCriticPolicy
    from .ppo import PPO, PPOConfig
    from .vec_normalize import VecNormalize
    env = EnvironmentFactory.create("CustomMyoReorientP1", num_envs=envs, seed=1)
    torch.manual_seed(0)
    pol = ActorCriticPolicy(env.obs_dim, env.act_dim, (256, 256), (256, 256), lstm_hidden_size=256, log_std_init=-2.0)
    algo = PPO(VecNormalize(env), pol, PPOConfig(n_steps=n_steps, batch_size=envs * n_steps // 8, n_epochs=4, learning_rate=2.5e-5))
    algo.collect_rollouts(); algo.train()                       # warm-up (captures the graphs)
    torch.cuda.synchronize()
    t0 = time.time(); tr = 0.0
    for _ in range(iters):
        t1 = time.time(); algo.collect_rollouts(); torch.cuda.synchronize(); tr += time.time() - t1
        algo.train()
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = iters * envs * n_steps
    out = {"config": "E: CustomMyoReorientP1, %d envs, LSTM-256 + MLP[256,256]" % envs,
           "env_steps_per_sec_rollout_plus_update": steps / dt, "env_steps_per_sec_rollout_only": steps / tr,
           "n_steps": n_steps, "epochs": 4, "dtype": "mixed", "recurrent_path": "fused" if algo._fused_rec is not None else "autograd"}
    if env_only_steps:                                          # physics alone (zero actions)
        act = torch.zeros((envs, env.act_dim), device=env.device)
        t2 = time.time()
        for _ in range(env_only_steps):
            env.step_tensor(act)
        torch.cuda.synchronize()
        out["env_steps_per_sec_env_only"] = env_only_steps * envs / (time.time() - t2)
    env.close()
    return out
