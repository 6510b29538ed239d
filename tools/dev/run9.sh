#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT
timeout 900 python3 -m pytest tests/test_step_parts.py -m gpu -x -q 2>&1 | tail -5
