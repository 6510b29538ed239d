import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from myochallenge_amd.envs.environment_factory import EnvironmentFactory
from myochallenge_amd.rl.sb3_zip import load_policy
class Ident:
    training = True
    def normalize_obs(self, o): return o
pol, _ = load_policy("tests/golden/phase1_final.zip")
N = 4096
mix = EnvironmentFactory.create("MixtureModelBaodingEnv", num_envs=N, seed=1, base_model_path=None, base_env_path=None, base_policy=pol, base_normalizer=Ident())
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.clamp(torch.randn((N, 39), device="cuda", generator=g) * 0.135, -1, 1) for _ in range(8)]
mix.reset_tensor()
torch.cuda.synchronize()
for t in range(10):
    n0 = mix.base_phase_launches
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = mix.step_tensor(acts[t % 8])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(t, "done", int(out[2].sum()), "phases", mix.base_phase_launches - n0, "ms", round(1e3 * dt, 2), "graph", mix._graph is not None)
# one phase timing
torch.cuda.synchronize(); t0 = time.perf_counter(); mix._run_phase(); torch.cuda.synchronize(); print("one phase ms", 1e3 * (time.perf_counter() - t0))
t0 = time.perf_counter(); mix._phase_body(); torch.cuda.synchronize(); print("one eager phase ms", 1e3 * (time.perf_counter() - t0))
