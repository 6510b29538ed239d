#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT
for l in p32 p64; do echo == $l; python3 tools/dev/gpu_mlp_prof.py tools/dev/lib_$l.so 2>&1 | grep -v "Warn\|warn\|amdgpu.ids"; done
