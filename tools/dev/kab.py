"""A/B timing of k_step across prebuilt library variants on ONE box (developer tool, run through gpurun):

    python tools/dev/kab.py [--envs 4096] [--dtype mixed] [--integrator euler] [--steps 60] [--rounds 2] libA.so libB.so ...

Each variant runs in a child process (the model lives in __constant__ memory of its own code object): 40 warm-up env steps
with N(0, 0.135) actions from a fixed seed, then `steps` timed steps (HIP events around every myo_batch_step launch).
Prints per variant: mean / min kernel ms per launch and a checksum of the final state — variants that claim "same
arithmetic" must print the same checksum.  Rounds alternate the variants (A B A B) so that clock drift shows."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(args):
    import hashlib
    import torch
    sys.path.insert(0, ROOT)
    from myochallenge_amd import native
    from myochallenge_amd.envs.config import make_task_cfg
    from myochallenge_amd.model import compile_model
    from myochallenge_amd.synth_hand import synthetic_hand, synthetic_hand_die
    lib = native.load(os.path.abspath(args.child.split("@")[0]))
    dev = torch.device("cuda:0")
    integ = {"euler": 0, "rk4": 1}[args.integrator]
    dtype = native.MYO_F64 if args.dtype == "f64" else native.MYO_MIXED
    N = args.envs
    if args.env == "reorient":
        from myochallenge_amd.envs.reorient import make_reorient_cfg
        cm = compile_model(synthetic_hand_die(), integrator=integ, unsupported_contacts="drop")
        tc = make_reorient_cfg("CustomMyoReorientP2", cm)
    else:
        cm = compile_model(synthetic_hand(), integrator=integ)
        tc = make_task_cfg("CustomMyoBaodingBallsP2" if args.env == "p2" else "CustomMyoBaodingBallsP1", cm)
    b = native.Batch(native.Model(cm, lib), tc, N, 0, 1, dtype)
    nobs = b.obs_dim
    obs = torch.zeros((N, nobs), dtype=torch.float32, device=dev)
    rew = torch.zeros(N, dtype=torch.float32, device=dev)
    done = torch.zeros(N, dtype=torch.uint8, device=dev)
    b.reset(None, obs)
    bad = torch.zeros(N, dtype=torch.uint8, device=dev)
    bad_total = torch.zeros(N, dtype=torch.int32, device=dev)
    b.set_bad_state_buffer(bad)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    acts = [torch.clamp(torch.randn((N, 39), device=dev, generator=g) * 0.135, -1, 1) for _ in range(16)]
    for t in range(args.warm):
        b.step(acts[t % 16], obs, rew, done)
    if args.tune:
        b.tune_wrap_order()
    torch.cuda.synchronize()
    b.enable_timing(True)
    for t in range(args.steps):
        b.step(acts[t % 16], obs, rew, done)
        bad_total += bad
    torch.cuda.synchronize()
    ms = b.kernel_ms()
    qp = torch.zeros((N, cm.size("nq")), dtype=torch.float64, device=dev)
    b.get_state(qp)
    torch.cuda.synchronize()
    h = hashlib.sha1(qp.cpu().numpy().tobytes()).hexdigest()[:12]
    nb = int(bad_total.sum().item())
    print(json.dumps({"ms": ms, "sum": h + ("" if nb == 0 else "+bad%d@%s" % (nb, ",".join(str(int(i)) for i in torch.nonzero(bad_total).flatten()[:4].tolist()))), "lds": b.lds_bytes}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--dtype", default="mixed")
    ap.add_argument("--integrator", default="euler")
    ap.add_argument("--env", default="p1", choices=["p1", "p2", "reorient"])
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warm", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--tune", type=int, default=0, help="1: myo_batch_tune_wrap_order after the warm-up steps")
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args)
    res = {l: [] for l in args.libs}
    for r in range(args.rounds):
        for l in args.libs:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", l, "--envs", str(args.envs), "--dtype", args.dtype,
                   "--integrator", args.integrator, "--env", args.env, "--steps", str(args.steps), "--warm", str(args.warm), "--tune", str(args.tune)]
            env = dict(os.environ)
            for kv in l.split("@")[1:]:           # lib.so@NAME=VALUE: the variant runs with that environment variable
                k, v = kv.split("=", 1); env[k] = v
            try:
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
                d = json.loads(out.stdout.strip().splitlines()[-1])
                res[l].append(d)
                print(f"round {r} {os.path.basename(l):40s} {d['ms']:.4f} ms  sum {d['sum']} lds {d['lds']}", flush=True)
            except Exception as exc:
                print(f"round {r} {os.path.basename(l):40s} FAILED {exc!r}\n{out.stderr[-800:] if 'out' in dir() else ''}", flush=True)
    print("---- summary (envs %d, %s, %s, %s)" % (args.envs, args.dtype, args.integrator, args.env))
    for l, v in res.items():
        if v:
            ms = [d["ms"] for d in v]
            print(f"{os.path.basename(l):40s} mean {sum(ms)/len(ms):.4f}  min {min(ms):.4f}  sum {v[0]['sum']}")


if __name__ == "__main__":
    main()
