#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gradient_tail or fused_rollout" 2>&1 | tail -12
