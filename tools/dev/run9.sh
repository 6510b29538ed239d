#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT; mkdir -p gpurun_out/r03
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 tools/dev/gpu_mlp_check.py 2>&1 | tail -5
