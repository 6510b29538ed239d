"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the kernel SOURCE (lane-serial -DMYO_EMU build; GPU ASan is not
available on the pool) and of the oracle, as part of the CPU suite: the sanitized libraries are built here and driven in a
CHILD process that has libasan / libubsan preloaded; any report makes the child exit non-zero.

Covers: a mixed-Euler episode slice, a mixed-RK4 slice, the die-reorient case (resets, object group), an fp64 slice, and the
C model loader on corrupt / mutated .mjb files (myo_model_load_mjb must not trust its input)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from myochallenge_amd import native
from myochallenge_amd.build import build_emu
import oracle.oracle as orc
orc._LIB = None
_real_build = orc.build
orc.build = lambda force=False: os.path.join(%(root)r, "oracle", "libmyo_oracle_asan.so")      # the sanitized checker
lib = native.load(build_emu(sanitize=True))
import parity_cases as pc
from myochallenge_amd.synth_hand import synthetic_hand
hand = synthetic_hand()
r = pc.episode_drift(lib, hand, native.MYO_MIXED, [(0.135, 1), (0.08, 2)], 25)
assert r["err_qpos_rel"].max() <= 1e-4
r = pc.episode_drift(lib, hand, native.MYO_MIXED, [(0.135, 1)], 12, integrator=1)
assert r["err_qpos_rel"].max() <= 1e-4
r = pc.episode_drift(lib, hand, native.MYO_F64, [(0.135, 3)], 15, env_name="CustomMyoBaodingBallsP2", resync=True)
assert r["err_qpos_rel"].max() <= 1e-9
pc.case_reorient(lib, native.MYO_MIXED, 1e-4, n=3, nsteps=14)
# the C model loader on corrupt files
import tempfile
from myochallenge_amd.mjb import dump_mjb
import test_mjb_and_model as tm
good = dump_mjb(hand)
d = tempfile.mkdtemp()
def load(raw):
    p = os.path.join(d, "m.mjb"); open(p, "wb").write(raw)
    return native.Model.from_mjb(p, lib)
assert load(good).size("nv") == 35
bad = list(tm.corrupt_mjb_variants(hand, good)) + [dump_mjb(m) for m in tm.corrupt_id_variants(hand)]
rng = np.random.RandomState(0)
for k in range(60):                      # random single-word mutations of the header / id arrays
    b = bytearray(good)
    off = int(rng.randint(16, 16 + 57 * 4 + 200)) if k %% 2 else int(rng.randint(len(good) - 70000, len(good) - 4)) & ~3
    b[off:off + 4] = int(rng.choice([-1, 0x7fffffff, 0x10000, -70000, 255])).to_bytes(4, "little", signed=True)
    bad.append(bytes(b))
rejected = 0
for raw in bad:
    try:
        m = load(raw)
        b = native.Batch(m, None, 1, 0, 0, native.MYO_F64)      # a mutation that still loads must also step cleanly
        b.physics_step(np.zeros((1, m.size("nu"))), 2)
        b.close()
    except native.MyoError:
        rejected += 1
assert rejected >= 25, rejected
print("sanitizer child ok", rejected, len(bad))
'''


def _runtime(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.slow
def test_kernel_source_and_oracle_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    from myochallenge_amd.build import build_emu
    build_emu(sanitize=True)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1")
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0 and "sanitizer child ok" in res.stdout, (res.returncode, res.stdout[-2000:], res.stderr[-4000:])
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-4000:]
