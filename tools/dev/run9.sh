#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT; mkdir -p gpurun_out/r03
timeout 600 python3 tools/train_demo.py --steps 100000000 --out gpurun_out/r03/train_demo_p1_100m.json 2>&1 | grep -v amdgpu.ids | tail -12
timeout 600 python3 tools/train_demo.py --env-name CustomMyoReorientP1 --steps 60000000 --out gpurun_out/r03/train_demo_reorient_60m.json 2>&1 | grep -v amdgpu.ids | tail -8
