#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants > /tmp/stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats_mid.csv \;
tail -1 /tmp/stats.log > $OUT/bench_line_under_rocprof_mid.json
