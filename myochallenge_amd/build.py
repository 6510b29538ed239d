"""Build recipes for the native pieces (all in-tree, so the .so files travel with gpurun).

* ``build_hip()``   hipcc --offload-arch=gfx950 -> myochallenge_amd/libmyobatch.so (the product)
* ``build_emu()``   g++ -DMYO_EMU csrc/myobatch_emu.cpp -> tests/emu/libmyobatch_emu.so (test tooling:
                    lane-serial emulation of the kernel source, optional sanitizers)
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["myobatch.hip", "myo_host.h", "myo_task.h", "myo_physics.h", "myo_model_dev.h", "wave.h", "myo_mjb.h", "mjb_layout.inc",
           "myo_ppo_mlp.h", "myo_sparse_ldl.h", "myo_arrow_chol.h", "myo_lstm_step.h", "myo_lstm_seq.h"]
HEADERS = [os.path.join(ROOT, "include", "myobatch.h"), os.path.join(ROOT, "include", "myo_model_blob.h")]
# the emulation build's own translation unit and backend (TEST TOOLING: nothing of it is compiled into libmyobatch.so, and it is not part
# of source_id(), the identity of the product's sources)
EMU_SOURCES = ["myobatch_emu.cpp", "emu_host.h"]


def reachable_includes(entry: str = "myobatch.hip") -> set:
    """Every file a quoted #include reaches from ``entry`` (absolute paths).  tests/test_config_and_abi.py checks
    that SOURCES + HEADERS cover this set, so _stale() and source_id() see every file the library is made of."""
    import re
    seen, todo = set(), [os.path.join(CSRC, entry)]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), re.M):
            todo.append(os.path.join(os.path.dirname(f), inc))
    return seen


# Code-generation flags of the product build (part of its identity: source_id() hashes them with the sources).
# -fno-hip-fp32-correctly-rounded-divide-sqrt: 2.5-ulp fp32 div/sqrt (fewer VALU instructions; the fp64 stepper is unaffected).
# (Measured and NOT used: -mllvm -amdgpu-sched-strategy=max-ilp makes k_step<double> 1.2 % faster on one box (2.003 -> 1.979 ms, same
#  checksums) — and lets leaf functions grow to 248 VGPRs, where their callee-saved registers no longer fit into AGPRs and go to
#  scratch memory: tendon_wrap_pass saves ten of them per call and the launch's write traffic goes from 48 to 119 MB.
#  max-memory-clause +0.3 %, iterative-minreg +9 %, -O2 +0.3 %, no post-RA scheduling +0.9 %; iterative-ilp crashes the compiler.)
HIP_CODEGEN_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-hip-fp32-correctly-rounded-divide-sqrt"]


def source_id() -> str:
    """Short hash of the native sources and the code-generation flags: compiled into the library (myo_version()) so that
    measured records (profiles/*_pmc.json) can be matched to the build they were taken on."""
    import hashlib
    h = hashlib.sha1()
    h.update(" ".join(HIP_CODEGEN_FLAGS).encode())
    for f in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def _stale(target: str, extra=()) -> bool:
    if not os.path.exists(target):
        return True
    if not extra:      # the product library carries the hash of the sources and flags it was built from: that, not file times, decides
        try:
            return (" build " + source_id()).encode() not in open(target, "rb").read()
        except OSError:
            return True
    t = os.path.getmtime(target)
    deps = [os.path.join(CSRC, s) for s in list(SOURCES) + list(extra)] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    out = os.path.join(HERE, "libmyobatch.so")
    if force or _stale(out):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not os.path.exists(hipcc):
            hipcc = "hipcc"
        # (the parity tests run against this exact build)
        cmd = [hipcc] + HIP_CODEGEN_FLAGS + ["-fPIC", "-shared", "-Wno-invalid-offsetof", '-DMYO_BUILD_ID="%s"' % source_id(),
                                             os.path.join(CSRC, "myobatch.hip"), "-o", out]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
    return out


def build_emu(force: bool = False, sanitize: bool = False) -> str:
    d = os.path.join(ROOT, "tests", "emu")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, "libmyobatch_emu_asan.so" if sanitize else "libmyobatch_emu.so")
    if force or _stale(out, EMU_SOURCES):
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-invalid-offsetof", "-DMYO_EMU", '-DMYO_BUILD_ID="%s"' % source_id(),
               os.path.join(CSRC, "myobatch_emu.cpp"), "-o", out]
        if sanitize:
            cmd[1:1] = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
        subprocess.check_call(cmd)
    return out
