#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd $ROOT; mkdir -p gpurun_out/r03
python3 bench.py > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err
tail -c 600 gpurun_out/r03/bench_final.err
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
