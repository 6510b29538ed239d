#!/bin/bash
# The counters VERDICT's targets are written in, for the k_step launches of bench.py (one --pmc pass per set): run through gpurun.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmcq
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 "$@" > /tmp/pmc.log 2>&1
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT/pmc.txt
done
cat $OUT/pmc.txt
