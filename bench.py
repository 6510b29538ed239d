#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched Baoding path (BASELINE.json metric) on N MI355X GPUs.

One "step" = one pass of the hot path over one batch: policy inference (MLP[256,256], bf16) ->
clip -> myo_batch_step (10 physics substeps + obs/reward/termination/auto-reset for every env of
the batch) -> VecNormalize -> rollout-buffer write; every --n-steps steps a full PPO update
(GAE, n_epochs x minibatches, grad all-reduce over RCCL when N>1) runs INSIDE the timed region.
`value` is the whole-job aggregate (all ranks) env-steps/s.

Workload (config.workload): BASELINE.json configs[1] — Baoding phase-1 config, 4096 batched
envs per GPU, PPO MLP[256,256] bf16.  The model is the labelled SYNTHETIC MyoHand stand-in
(myochallenge_amd/synth_hand.py) because the reference's myo_hand_baoding.mjb is a stripped
blob; the integrator is the model's own option (Euler, as in the only decodable MyoSuite
model); --integrator rk4 runs the RK4 variant north_star mentions.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_F64_STATE = 2508  # algorithmic HBM bytes per env-step, fp64 state records


def cpu_baseline(seconds: float, all_cores: bool):
    """Times the oracle (scalar fp64 C port of the same env step) on the host cores.
    The reference's own CPU path (MuJoCo + MyoSuite + SubprocVecEnv) cannot run here."""
    import numpy as np

    def worker(sec, seed, q=None):
        from myochallenge_amd.envs.config import task_ids
        from myochallenge_amd.model import compile_model
        from myochallenge_amd.synth_hand import synthetic_hand
        from oracle.oracle import BaodingState, OracleData, OracleModel, baoding_step, make_cfg
        cm = compile_model(synthetic_hand())
        d = OracleData(OracleModel(cm.to_blob()))
        cfg = make_cfg(task_ids(cm))
        rng = np.random.RandomState(seed)

        def reset():
            d.reset()
            d.qpos[0] = -1.57
            st = BaodingState()
            st.which_task, st.counter = 2, 0
            st.start_angle[0], st.start_angle[1] = 3 * np.pi / 4, -np.pi / 4
            st.x_radius, st.y_radius, st.time_period = 0.025, 0.028, 5.0
            return st
        st, n, ep, t0 = reset(), 0, 0, time.time()
        while time.time() - t0 < sec:
            a = np.clip(rng.normal(0, 0.135, 39), -1, 1).astype(np.float32)
            _, c = baoding_step(d, cfg, st, a)
            n += 1
            ep += 1
            if c[6] or ep >= 200 or d.bad:
                st, ep = reset(), 0
        rate = n / (time.time() - t0)
        if q is not None:
            q.put(rate)
        return rate

    one = worker(seconds, 0)
    out = {"value": one, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{seconds:.0f} s of one env, synthetic MyoHand Baoding P1, N(0,0.135) actions, 200-step episodes"}
    if all_cores:
        import multiprocessing as mp
        nc = os.cpu_count() or 1
        ctx = mp.get_context("fork")
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(seconds, 100 + i, q)) for i in range(nc)]
        [p.start() for p in ps]
        rates = [q.get() for _ in ps]
        [p.join() for p in ps]
        out["all_cores"] = {"value": float(sum(rates)), "cores": nc}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--integrator", default="model", choices=["model", "euler", "rk4"])
    ap.add_argument("--n-steps", type=int, default=64, help="rollout length between PPO updates")
    ap.add_argument("--n-epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=16384, help="minibatch (default: 1/16 of the default rollout, the reference's ratio: 4096 of 65536)")
    ap.add_argument("--env-name", default="CustomMyoBaodingBallsP1")
    ap.add_argument("--lstm-hidden", type=int, default=0, help="recurrent policy (not the headline config): LSTM of this width for "
                    "actor and critic; 128 with --net-arch '' is the architecture of trained_models/phase_1/phase1_final.zip")
    ap.add_argument("--net-arch", default="256,256", help="MLP widths after the (optional) LSTM")
    ap.add_argument("--no-ppo", action="store_true", help="rollout only (reported as invalid for the headline)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the stepper has no CPU execution path")
    local_rank %= max(1, torch.cuda.device_count())     # (lets a 1-GPU box exercise the N>1 path with gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MYO_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; gloo only for single-GPU tests of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from myochallenge_amd.envs.environment_factory import EnvironmentFactory
    from myochallenge_amd.rl.policy import ActorCriticPolicy
    from myochallenge_amd.rl.ppo import PPO, PPOConfig
    from myochallenge_amd.rl.vec_normalize import VecNormalize

    # every timed region must contain at least one full PPO update: if K is smaller than the configured rollout
    # the rollout shrinks to K steps while the minibatch stays (close to) the configured size, so the optimizer
    # work per env step — and the GEMM shapes — are those of the headline configuration for any K
    if args.steps < args.n_steps:
        args.n_steps = max(1, args.steps)
    rollout = args.n_steps * args.envs
    n_mb = max(1, rollout // args.batch_size)
    args.batch_size = max(1, (rollout // n_mb) // 128 * 128) if rollout >= 128 else rollout
    integ = None if args.integrator == "model" else args.integrator
    env = EnvironmentFactory.create(args.env_name, num_envs=args.envs, device=local_rank, seed=1234 + rank,
                                    dtype=args.dtype, integrator=integ)
    integ_name = {0: "Euler", 1: "RK4"}[env._model.size("integrator")]
    venv = VecNormalize(env, gamma=0.99)
    torch.manual_seed(0)   # identical initial weights on every rank
    arch = tuple(int(x) for x in args.net_arch.split(",") if x.strip())
    policy = ActorCriticPolicy(env.obs_dim, env.act_dim, arch, arch, lstm_hidden_size=args.lstm_hidden or None, log_std_init=-2.0)
    cfg = PPOConfig(n_steps=args.n_steps, batch_size=args.batch_size, n_epochs=args.n_epochs, learning_rate=2.5e-4,
                    clip_range=0.2, ent_coef=2.5e-4, vf_coef=0.5, gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, bf16=True)
    algo = PPO(venv, policy, cfg, seed=rank)

    # one "step" of the bench = one rollout step (PPO.rollout_step: hipGraph(policy) -> myo_batch_step
    # -> hipGraph(normaliser, bootstrap, buffer write)); the PPO update fires every n_steps steps
    state = {"t": 0}

    def one_step():
        algo.rollout_step()
        state["t"] += 1
        if state["t"] == cfg.n_steps:
            state["t"] = 0
            algo.finish_rollout()
            if not args.no_ppo:
                algo.train()

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # untimed set-up: one full rollout + update so that the hipGraph of the optimizer step is
    # captured before anything is measured; then the W warm-up steps of the contract
    if not args.no_ppo:
        for _ in range(cfg.n_steps):
            one_step()
    for _ in range(args.warmup):
        one_step()
    fence()
    env.batch.enable_timing(True)
    upd0 = algo.n_updates
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = env.batch.kernel_ms()
    env.batch.enable_timing(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    total_envs = args.envs * world
    value = total_envs * args.steps / elapsed

    pol_name = (f"LSTM-{args.lstm_hidden} + " if args.lstm_hidden else "") + f"MLP[{args.net_arch}]"
    if rank == 0:
        alg_bytes = ALG_BYTES_F64_STATE * args.envs
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
        traffic, valu = None, None
        try:   # PMC figures are collected off-line with rocprofv3 (profiles/README.md) for the default workload
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            if args.envs == pmc["envs_per_launch"] and args.dtype == "f32" and integ_name == "Euler":
                traffic = pmc["hbm_bytes_per_launch"]
                lanes_per_s = pmc["SQ_INSTS_VALU"] * 64 / (kernel_ms * 1e-3)
                valu = {"valu_wave_instructions_per_launch": pmc["SQ_INSTS_VALU"],
                        "achieved_lane_ops_per_s": lanes_per_s, "peak_lane_ops_per_s": 78.65e12,
                        "frac_of_valu_issue_peak": lanes_per_s / 78.65e12,
                        "note": "peak = 157.3 TFLOP/s fp32 vector / 2 (one FMA = 2 flop) = lane-instructions/s"}
        except Exception:
            pass
        out = {
            "metric": "env-steps/sec (Baoding, 4096 envs)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (synthetic MyoHand-shaped model, random-init policy)",
            "config": {"workload": f"Baoding phase-1 config ({args.env_name}), {args.envs} batched envs per GPU, "
                                   f"PPO {pol_name} bf16, frame_skip 10, {integ_name} integrator (model option)",
                       "envs_per_gpu": args.envs, "global_envs": total_envs, "integrator": integ_name,
                       "ppo": "rollout-only" if args.no_ppo else
                       f"n_steps={cfg.n_steps}, batch={cfg.batch_size}, epochs={cfg.n_epochs}, update inside timed region",
                       "parallelism": f"env-sharded x{world}, 1 RCCL grad all-reduce per optimizer step"},
            "ppo_optimizer_steps_per_sec": (algo.n_updates - upd0) / elapsed,
            "env_kernel_ms": kernel_ms,
            "env_kernel_only_steps_per_sec_per_gpu": args.envs / (kernel_ms * 1e-3) if kernel_ms > 0 else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": (achieved / 8000.0) if achieved else None, "traffic": traffic, "valu": valu,
                         "kernel": "k_step<%s>" % ("float" if args.dtype == "f32" else "double"),
                         "note": "algorithmic bytes = 2508 B/env-step x envs per launch; the kernel is a long "
                                 "dependent chain of small vector ops (VALU/latency-bound), so the HBM fraction "
                                 "is tiny by construction — see DESIGN.md for the VALU-side accounting"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, all_cores=True)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    env.close()


if __name__ == "__main__":
    main()
