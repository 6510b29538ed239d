"""Developer check: LOCAL (per env step, re-synchronised) error of a stepper build on the 16 seeded streams; prints the worst steps.
    python tools/dev/gpu_local_rk4.py <lib.so> <euler|rk4> <mixed|f64> [nsteps]"""
import os, sys
import numpy as np
import torch  # before the library: one HIP runtime per process
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from myochallenge_amd import native
from myochallenge_amd.synth_hand import synthetic_hand
lib = native.load(os.path.abspath(sys.argv[1]))
integ = 1 if sys.argv[2] == "rk4" else 0
dt = native.MYO_MIXED if sys.argv[3] == "mixed" else native.MYO_F64
n = int(sys.argv[4]) if len(sys.argv) > 4 else 60
streams = [(sg, seed) for sg in (0.08, 0.135) for seed in range(8)]
r = pc.local_error(lib, synthetic_hand(), dt, streams, n, integrator=integ)
np.set_printoptions(linewidth=220, precision=2)
print(sys.argv[1:], "max per stream", r["err_qpos_rel"].max(1))
print("stream 5:", r["err_qpos_rel"][5][:40])
print("disagreements", r["done_disagreements"])
