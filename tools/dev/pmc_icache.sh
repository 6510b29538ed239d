#!/bin/bash
# Instruction-fetch / scalar-cache counters of k_step (is the code footprint — ~150 KB of hot leaves + kernel body against a 64 KB
# instruction cache per two CUs — visible?): run through gpurun from the repo root.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/icache
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
: > $OUT/icache.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 "$@" > /tmp/pmc.log 2>&1
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT/icache.txt
  tail -2 /tmp/pmc.log | grep -i "error\|not" >> $OUT/icache.txt
done
cat $OUT/icache.txt
