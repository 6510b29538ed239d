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

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides
and taken as the max over ranks; blocks repeat until --min-seconds (default 2 s) of timed wall have passed, so
that the GPU is busy long enough for outside telemetry; `ms_per_step` / `value` are the MEDIAN block
(`blocks`, `timed_seconds`, `ms_per_step_min/max` say what was seen).  `variants` (rank 0, N = 1) carries the
same measurement for the mixed-precision stepper and for the RK4 integrator, `config_C_p2_8192` (BASELINE config C: phase 2, 8192 envs) and
`reorient_p2_mlp` (the die env with the MLP policy) on the fp64 stepper, plus `config_E_lstm256`: BASELINE config E (die reorient,
4096 envs, the reference's recurrent LSTM-256 policy class, rollout + update; myochallenge_amd/rl/bench_reorient_lstm.py).  Since round 4 the headline is the all-fp64 stepper
(`--dtype f64`, the reference's own arithmetic); `--dtype mixed` is the faster variant whose per-step error is bounded in
tests/test_gpu_parity.py::test_local_error_of_the_steppers.

Launch:  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: bench.py starts the N ranks itself)
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
    """Times the oracle (scalar fp64 C port of the same env step) on the host cores, and counts its flops.
    The reference's own CPU path (MuJoCo + MyoSuite + SubprocVecEnv) cannot run here (bench/ref_subproc.py is the
    hook for a machine that has it)."""
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
    out["flops"] = count_flops()
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


def count_flops(worker=None, env_steps=600):
    """Algorithmic flops per env step, COUNTED by the instrumented oracle over whole episodes of the
    bench workload (oracle/myo_oracle.c, FL()): 1 per add / sub / mul / div / sqrt / transcendental of the algorithm
    as restated there (dense constraint Jacobian, lower triangle of J'DJ, tree-sparse M v).  Runs in a child process
    on the counting build of the oracle (libmyo_oracle_flops.so); the timed baseline uses the build without it."""
    import subprocess
    code = ("import json, sys; sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench._count_flops_here(%d)))" % (ROOT, env_steps))
    env = dict(os.environ, MYO_ORACLE_FLOPS="1")
    try:
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, check=True).stdout
        return json.loads(out.strip().splitlines()[-1])
    except Exception as exc:
        return {"error": repr(exc)}


def _count_flops_here(env_steps):
    import numpy as np
    from myochallenge_amd.envs.config import task_ids
    from myochallenge_amd.model import compile_model
    from myochallenge_amd.synth_hand import synthetic_hand
    from oracle import oracle as orc
    from oracle.oracle import BaodingState, OracleData, OracleModel, baoding_step, make_cfg
    cm = compile_model(synthetic_hand())
    d = OracleData(OracleModel(cm.to_blob()))
    cfg = make_cfg(task_ids(cm))
    rng = np.random.RandomState(1)

    def reset():
        d.reset()
        d.qpos[0] = -1.57
        st = BaodingState()
        st.which_task, st.counter = 2, 0
        st.start_angle[0], st.start_angle[1] = 3 * np.pi / 4, -np.pi / 4
        st.x_radius, st.y_radius, st.time_period = 0.025, 0.028, 5.0
        return st
    st, ep = reset(), 0
    orc.flops(reset=True)
    for _ in range(env_steps):
        a = np.clip(rng.normal(0, 0.135, 39), -1, 1).astype(np.float32)
        _, c = baoding_step(d, cfg, st, a)
        ep += 1
        if c[6] or ep >= 200 or d.bad:
            st, ep = reset(), 0
    fl = orc.flops(reset=True)
    tot = sum(fl.values())
    return {"per_env_step": tot / env_steps, "env_steps_counted": env_steps,
            "stage_share": {k: round(v / max(1.0, tot), 4) for k, v in fl.items()}}


FLOPS_RECORD = os.path.join(ROOT, "profiles", "r02_flops.json")     # the same count, committed (used when the CPU leg is skipped)
PMC_RECORD = os.path.join(ROOT, "profiles", "r06_pmc.json")         # rocprofv3 --pmc passes over the default command (tools/profile_round.sh)
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9                           # 78.6e12 lane-instructions/s: 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md)


def trajectory_parity():
    """A POINTER to the whole-episode drift records the -m gpu parity tests wrote (HIP vs oracle, 16 action streams), not a result of
    this run: per stepper / config [worst stream's max error, streams within north_star's 1e-4, streams], read from the newest committed
    profiles/<round>_drift_<name>.json (the replayed records used to take half of the line)."""
    out = {"note": "committed records of tests/test_gpu_parity.py::test_episode_trajectory_* / tests/test_reorient.py (HIP vs oracle, "
                   "err = max|qpos - qpos_oracle| / max|qpos_oracle| per env step); NOT measured in this run",
           "records": "profiles/<round>_drift_<name>.json, profiles/<round>_local_error_<name>.json",
           "worst_within1e-4_streams": {}}
    for key, name in (("f64_euler", "f64"), ("mixed_euler", "mixed"), ("f64_rk4", "rk4_f64"), ("mixed_rk4", "rk4_mixed"),
                      ("configC_f64", "configC_f64"), ("configC_mixed", "configC_mixed"), ("configE_f64", "configE_f64")):
        path = next((q for q in (os.path.join(ROOT, "profiles", "%s_drift_%s.json" % (rr, name)) for rr in ("r06", "r05", "r04", "r03")) if os.path.exists(q)), "")
        try:
            mq = json.load(open(path))["max_err_qpos_rel"]
            out["worst_within1e-4_streams"][key] = [float("%.3g" % max(mq)), sum(1 for v in mq if v <= 1e-4), len(mq), os.path.basename(path)[:3]]
        except Exception:
            out["worst_within1e-4_streams"][key] = None
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torchrun with the
    rendezvous on 127.0.0.1) as a CHILD process and relay its output and exit code.  Called before torch is imported, so
    this parent never initialises the GPU (a process that has must not be replaced or forked on this pool)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--dtype", default="f64", choices=["mixed", "f64", "f32"],
                    help="stepper arithmetic: f64 = everything fp64, the reference's own arithmetic (MuJoCo), the headline since round 4; "
                         "mixed = fp64 state / kinematic chain / contact distances / tendon lengths, fp32 dynamics (f32 is round 1's name for it)")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step block until this much timed wall has passed")
    ap.add_argument("--no-variants", action="store_true", help="skip the f64 / RK4 variant measurements (rank 0, N = 1 only)")
    ap.add_argument("--integrator", default="model", choices=["model", "euler", "rk4"])
    ap.add_argument("--n-steps", type=int, default=64, help="rollout length between PPO updates")
    ap.add_argument("--n-epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=16384, help="minibatch (default: 1/16 of the default rollout, the reference's ratio: 4096 of 65536)")
    ap.add_argument("--env-name", default="CustomMyoBaodingBallsP1")
    ap.add_argument("--lstm-hidden", type=int, default=0, help="recurrent policy (not the headline config): LSTM of this width for "
                    "actor and critic; 128 with --net-arch '' is the architecture of trained_models/phase_1/phase1_final.zip")
    ap.add_argument("--net-arch", default="256,256", help="MLP widths after the (optional) LSTM")
    ap.add_argument("--no-ppo", action="store_true", help="rollout only (reported as invalid for the headline)")
    ap.add_argument("--normalizer-sync", default="step", choices=["step", "rollout", "none"],
                    help="N > 1 ranks: VecNormalize statistics exchanged at every env step (one small all-reduce between the two rollout "
                         "graphs; the reference's single VecNormalize over all envs, exactly), once per rollout, or not at all")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

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

    # the library that runs must be the one the tree's sources produce: a stale .so would be benchmarked silently otherwise
    from myochallenge_amd import native as _native
    from myochallenge_amd.build import source_id as _source_id
    lib_build, src_build = _native.load().build_id, _source_id()
    if lib_build != src_build and not os.environ.get("MYO_ALLOW_STALE_LIB"):
        raise SystemExit(f"bench.py: libmyobatch.so is build {lib_build} but the sources are {src_build}: run __graft_entry__.build() "
                         "(MYO_ALLOW_STALE_LIB=1 overrides, for A/B runs of library variants)")

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(dtype, integrator, min_seconds, steps, warmup, env_name=None, envs=None):
        """One configuration: set-up, W warm-up steps, then blocks of exactly `steps` steps until `min_seconds`."""
        env_name, envs = env_name or args.env_name, envs or args.envs
        from myochallenge_amd.envs.environment_factory import EnvironmentFactory
        from myochallenge_amd.rl.policy import ActorCriticPolicy
        from myochallenge_amd.rl.ppo import PPO, PPOConfig
        from myochallenge_amd.rl.vec_normalize import VecNormalize
        # every timed block must contain at least one full PPO update: if K is smaller than the configured rollout
        # the rollout shrinks to K steps while the minibatch stays (close to) the configured size, so the optimizer
        # work per env step — and the GEMM shapes — are those of the headline configuration for any K
        n_steps = min(args.n_steps, max(1, steps))
        rollout = n_steps * envs
        n_mb = max(1, rollout // args.batch_size)
        batch_size = max(1, (rollout // n_mb) // 128 * 128) if rollout >= 128 else rollout
        integ = None if integrator == "model" else integrator
        env = EnvironmentFactory.create(env_name, num_envs=envs, device=local_rank, seed=1234 + rank,
                                        dtype=dtype, integrator=integ)
        integ_name = {0: "Euler", 1: "RK4"}[env._model.size("integrator")]
        venv = VecNormalize(env, gamma=0.99, sync_ranks=args.normalizer_sync)
        torch.manual_seed(0)   # identical initial weights on every rank
        arch = tuple(int(x) for x in args.net_arch.split(",") if x.strip())
        policy = ActorCriticPolicy(env.obs_dim, env.act_dim, arch, arch, lstm_hidden_size=args.lstm_hidden or None, log_std_init=-2.0)
        cfg = PPOConfig(n_steps=n_steps, batch_size=batch_size, n_epochs=args.n_epochs, learning_rate=2.5e-4,
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

        # untimed set-up: one full rollout + update so that the hipGraph of the optimizer step is
        # captured before anything is measured; then the W warm-up steps of the contract
        if not args.no_ppo:
            for _ in range(cfg.n_steps):
                one_step()
        for _ in range(warmup):
            one_step()
        fence()
        env.batch.enable_timing(True)
        upd0 = algo.n_updates
        blocks, total, skew = [], 0.0, []
        while True:
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            if world > 1:
                torch.cuda.synchronize(dev)          # this rank's own GPU work is done ...
            own = time.perf_counter() - t0           # ... this long after the block started (before the closing barrier: the per-rank skew)
            fence()
            el = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el, own, -own], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt[0])
                skew.append((-float(tt[2]), float(tt[1])))       # (fastest, slowest) rank of the block
            blocks.append(el)
            total += el
            if total >= min_seconds or len(blocks) >= 10000:      # every rank sees the same (all-reduced) times
                break
        kernel_ms = env.batch.kernel_ms()
        env.batch.enable_timing(False)
        health = env.batch.health()                  # hand-off protocol errors, substeps that dropped contacts beyond the scratch's capacity
        replicas_identical, replica_spread = None, None
        if world > 1:      # data-parallel replicas must hold the same parameters after the timed updates (fp64 sum of |p|, min == max over ranks)
            cs = torch.stack([p.detach().double().abs().sum() for p in policy.parameters()]).sum().reshape(1)
            lo, hi = cs.clone(), cs.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            replicas_identical = bool(float(lo[0]) == float(hi[0]))
            replica_spread = float(hi[0]) - float(lo[0])
        blocks.sort()
        med = blocks[len(blocks) // 2] if len(blocks) % 2 else 0.5 * (blocks[len(blocks) // 2 - 1] + blocks[len(blocks) // 2])
        res = {"dtype": dtype, "integrator": integ_name, "value": envs * world * steps / med, "ms_per_step": 1e3 * med / steps,
               "ms_per_step_min": 1e3 * blocks[0] / steps, "ms_per_step_max": 1e3 * blocks[-1] / steps, "blocks": len(blocks),
               "timed_seconds": total, "env_kernel_ms": kernel_ms, "optimizer_steps_per_sec": (algo.n_updates - upd0) / total,
               "n_steps": cfg.n_steps, "batch_size": cfg.batch_size, "n_epochs": cfg.n_epochs, "lds_bytes": env.batch.lds_bytes, "health": health,
               "replicas_identical": replicas_identical, "replica_checksum_spread": replica_spread, "normalizer_sync": args.normalizer_sync if world > 1 else None,
               "graph_allreduce": bool(getattr(algo, "_allreduce_in_graph", False))}
        if world > 1 and getattr(algo, "vn_allreduce_calls", 0):      # host wall of the per-step eager normaliser all-reduce (rl/ppo.py rollout_step)
            res["normalizer_allreduce_ms_per_step"] = 1e3 * algo.vn_allreduce_seconds / algo.vn_allreduce_calls
            res["normalizer_allreduce_calls"] = int(algo.vn_allreduce_calls)
        if skew:      # per-rank spread of the block times (own work, before the barrier): min / max over ranks, median block
            sk = sorted(skew, key=lambda x: x[1])[len(skew) // 2]
            res["rank_block_seconds_min_max"] = [sk[0], sk[1]]
        env.close()
        del algo, venv, env, policy
        torch.cuda.empty_cache()
        return res

    dtype = "mixed" if args.dtype == "f32" else args.dtype
    main_res = measure(dtype, args.integrator, args.min_seconds, args.steps, args.warmup)
    variants = {}
    if rank == 0 and world == 1 and not args.no_variants and not args.no_ppo and not args.lstm_hidden:
        # the other steppers of the same workload, so that the driver's run records them too (short blocks: ~1 s each)
        other = "mixed" if dtype == "f64" else "f64"
        for name, (dt, integ) in {other: (other, args.integrator), "rk4": (dtype, "rk4"), "rk4_" + other: (other, "rk4")}.items():
            if (dt, integ) == (dtype, args.integrator):
                continue
            try:
                r = measure(dt, integ, min(1.0, args.min_seconds), min(args.steps, 64), min(args.warmup, 16))
                variants[name] = {k: r[k] for k in ("dtype", "integrator", "value", "ms_per_step", "env_kernel_ms", "blocks", "timed_seconds", "lds_bytes", "health")}
            except Exception as exc:      # a variant must never take the headline line down
                variants[name] = {"error": repr(exc)}
        if args.envs == 4096 and args.env_name == "CustomMyoBaodingBallsP1":
            # BASELINE config C (phase 2, randomised, 8192 envs) and the die-reorient env with the MLP policy, fp64 — short blocks too
            for name, (en, ne) in {"config_C_p2_8192": ("CustomMyoBaodingBallsP2", 8192), "reorient_p2_mlp": ("CustomMyoReorientP2", 4096)}.items():
                try:
                    r = measure("f64", "model", min(1.0, args.min_seconds), min(args.steps, 64), min(args.warmup, 16), env_name=en, envs=ne)
                    variants[name] = dict({k: r[k] for k in ("dtype", "integrator", "value", "ms_per_step", "env_kernel_ms", "blocks", "timed_seconds",
                                                              "lds_bytes", "health")}, env_name=en, envs=ne)
                except Exception as exc:      # noqa: BLE001
                    variants[name] = {"error": repr(exc)}
            # BASELINE config E (die reorient, LSTM-256 + MLP[256,256], the reference's RecurrentPPO policy class) on the same box
            try:
                from myochallenge_amd.rl.bench_reorient_lstm import run as run_config_e
                # the reference's own settings (n_steps 128, n_epochs 10, fp64 physics: src/main_reorient.py:53-71) ...
                variants["config_E_lstm256"] = run_config_e(4096, 128, 2, env_only_steps=0, reference_settings=True)
            except Exception as exc:      # noqa: BLE001
                variants["config_E_lstm256"] = {"error": repr(exc)}
            try:                          # ... and the light setting the earlier rounds quoted (32-step rollouts, 4 epochs) — on the fp64 stepper
                # since round 6: the mixed stepper is not offered for the die (its local error on edge active sets is 3e-5 per env step,
                # DESIGN.md §4)
                variants["config_E_lstm256_light"] = run_config_e(4096, 32, 2, env_only_steps=0, dtype="f64")
            except Exception as exc:      # noqa: BLE001
                variants["config_E_lstm256_light"] = {"error": repr(exc)}

    pol_name = (f"LSTM-{args.lstm_hidden} + " if args.lstm_hidden else "") + f"MLP[{args.net_arch}]"
    if rank == 0:
        kernel_ms, integ_name = main_res["env_kernel_ms"], main_res["integrator"]
        alg_bytes = ALG_BYTES_F64_STATE * args.envs
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(args.cpu_seconds, all_cores=True)
        flops = (cpu or {}).get("flops")
        if flops is None:
            try:
                flops = json.load(open(FLOPS_RECORD))
            except Exception:
                flops = None
        fl_roof = None
        if flops and kernel_ms > 0:
            per_step = flops["per_env_step"] * (4.0 if integ_name == "RK4" and not flops.get("rk4") else 1.0)
            peak = 78.6e12 if dtype == "f64" else 157.3e12      # gfx950 vector peak, fp64 / fp32 (MI355X_MICROARCH.md)
            ach = per_step * args.envs / (kernel_ms * 1e-3)
            fl_roof = {"per_env_step": per_step, "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
                       "stage_share": flops.get("stage_share"),
                       "note": "flops COUNTED by the instrumented oracle over whole episodes of this workload (x4 for RK4); "
                               "peak = vector (non-MFMA) peak of the arithmetic type; the mixed stepper runs part of them in fp64"}
        # PMC figures (HBM traffic, VALU instruction count) cannot be collected from inside this process: they come from
        # separate rocprofv3 --pmc passes over THIS command (tools/profile_round.sh -> profiles/<round>_pmc.json).  They
        # are attached only when that record was taken on the library build that is running now (source hash compiled
        # into myo_version()), for the same workload; otherwise the fields stay null and counters_source says why.
        traffic, valu, counters_source = None, None, None
        try:
            from myochallenge_amd import native
            build_id = native.load().build_id
            pmc = json.load(open(PMC_RECORD))
            same = (args.envs == pmc["envs_per_launch"] and dtype == pmc.get("dtype", "mixed") and integ_name == "Euler"
                    and "Baoding" in args.env_name and args.env_name.endswith("P1"))
            if not same:
                counters_source = "none: %s holds the default workload only" % os.path.relpath(PMC_RECORD, ROOT)
            elif pmc.get("build_id") != build_id:
                counters_source = "none: %s was taken on build %s, this library is build %s" % (
                    os.path.relpath(PMC_RECORD, ROOT), pmc.get("build_id"), build_id)
            else:
                traffic = pmc["hbm_bytes_per_launch"]
                lanes_per_s = pmc["SQ_INSTS_VALU"] * 64 / (kernel_ms * 1e-3)
                valu = {"valu_wave_instructions_per_launch": pmc["SQ_INSTS_VALU"], "achieved_lane_ops_per_s": lanes_per_s,
                        "lane_utilisation": pmc.get("lane_utilisation")}
                counters_source = "%s (offline rocprofv3 --pmc passes over this command, library build %s; replayed here, " \
                                  "divided by the kernel time measured in this run)" % (os.path.relpath(PMC_RECORD, ROOT), build_id)
        except Exception as exc:
            counters_source = "none: %r" % (exc,)
        kname = "k_step<%s>" % ("double" if dtype == "f64" else "float")
        out = {
            "metric": "env-steps/sec (Baoding, 4096 envs)" if "Baoding" in args.env_name else f"env-steps/sec ({args.env_name}, {args.envs} envs)", "value": main_res["value"], "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "rccl_ranks": world, "dist_backend": (dist.get_backend() if world > 1 else None),
            "dtype": {"mixed": "mixed: f64 state, kinematic chain, contact distances, tendon lengths; f32 dynamics", "f64": "f64"}[dtype],
            "data": "synthetic (synthetic MyoHand-shaped model, random-init policy)",
            "config": {"workload": f"{'Die-reorient' if 'Reorient' in args.env_name else 'Baoding phase-1' if args.env_name.endswith('P1') else 'Baoding phase-2'} config "
                                   f"({args.env_name}), {args.envs} batched envs per GPU, "
                                   f"PPO {pol_name} bf16, frame_skip {5 if 'Reorient' in args.env_name else 10}, {integ_name} integrator"
                                   f"{' (model option)' if args.integrator == 'model' else ''}",
                       "envs_per_gpu": args.envs, "global_envs": args.envs * world, "integrator": integ_name,
                       "ppo": "rollout-only" if args.no_ppo else
                       f"n_steps={main_res['n_steps']}, batch={main_res['batch_size']}, epochs={main_res['n_epochs']}, update inside timed region",
                       "parallelism": f"env-sharded x{world}, 1 RCCL grad all-reduce per optimizer step"},
            "blocks": main_res["blocks"], "timed_seconds": main_res["timed_seconds"],
            "ms_per_step_min": main_res["ms_per_step_min"], "ms_per_step_max": main_res["ms_per_step_max"],
            "trajectory_parity": trajectory_parity(),
            "ppo_optimizer_steps_per_sec": main_res["optimizer_steps_per_sec"],
            "env_kernel_ms": kernel_ms,
            "env_kernel_only_steps_per_sec_per_gpu": args.envs / (kernel_ms * 1e-3) if kernel_ms > 0 else None,
            # bound = VALU issue / latency: achieved = vector-ALU lane-instructions per second of the k_step
            # launches (PMC SQ_INSTS_VALU x 64 lanes / kernel time), peak = the chip's VALU issue peak; `hbm` and `flops`
            # are the two other ways of pricing the same launches.
            # the kernel is bound by the vector ALU (a long dependent chain of small fp64 vector ops), priced as COUNTED
            # algorithmic flops per launch / measured kernel time against the vector peak of the arithmetic type.  `valu_issue` is the other
            # reading of the same pipe — issue slots taken (PMC), which counts masked lanes and prices fp64 like fp32 — `hbm` the bytes.
            "roofline": {"bound": "valu",
                         "achieved": fl_roof["achieved"] if fl_roof else None, "peak": fl_roof["peak"] if fl_roof else None,
                         "unit": "TFLOP/s", "frac": fl_roof["frac"] if fl_roof else None,
                         "traffic": traffic, "counters_source": counters_source,
                         "valu_issue": {"achieved": (valu["achieved_lane_ops_per_s"] / 1e12) if valu else None, "peak": VALU_PEAK_LANE_OPS / 1e12,
                                        "unit": "Tlane-op/s", "frac": (valu["achieved_lane_ops_per_s"] / VALU_PEAK_LANE_OPS) if valu else None,
                                        "counters": valu},
                         "hbm": {"achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": (achieved / 8000.0) if achieved else None,
                                 "algorithmic_bytes_per_launch": alg_bytes},
                         "flops": fl_roof, "kernel": kname, "kernel_ms": kernel_ms, "lds_bytes_per_env": main_res["lds_bytes"],
                         "note": "one env per wavefront; the launch is a long dependent chain of small vector ops, so the bound is "
                                 "vector-ALU issue + LDS / instruction latency, not HBM (algorithmic bytes = 2508 B/env-step x envs "
                                 "per launch) and not MFMA; kernel_ms is measured live with HIP events on the launch stream"},
            "variants": variants,
            "health": main_res["health"],
            "library_build": lib_build,
        }
        if world > 1:
            out["replicas_identical"] = main_res["replicas_identical"]
            out["replica_checksum_spread"] = main_res["replica_checksum_spread"]
            out["config"]["normalizer_sync"] = main_res["normalizer_sync"]
            out["config"]["graph_allreduce"] = main_res["graph_allreduce"]
        if main_res.get("normalizer_allreduce_ms_per_step") is not None:
            out["normalizer_allreduce_ms_per_step"] = main_res["normalizer_allreduce_ms_per_step"]      # host wall per env step, rank 0 (whole run incl. warm-up)
        if main_res.get("rank_block_seconds_min_max"):
            out["rank_block_seconds_min_max"] = main_res["rank_block_seconds_min_max"]
        if any(main_res["health"].values()):
            print("bench.py: batch health counters are non-zero: %r" % (main_res["health"],), file=sys.stderr)
        flagged = {k: v["health"] for k, v in (out.get("variants") or {}).items() if isinstance(v, dict) and isinstance(v.get("health"), dict) and any(v["health"].values())}
        if flagged:      # (a variant's dropped contacts / protocol errors must not travel silently inside the line)
            out["variants_with_nonzero_health"] = sorted(flagged)
            print("bench.py: variants with non-zero batch health counters: %r" % (flagged,), file=sys.stderr)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
