#!/bin/bash
# Instruction-class mix of k_step (dynamic, per launch): run through gpurun from the repo root.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/mix
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INSTS_[A-Z0-9_]*\|SQ_ACTIVE_INST[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*" $OUT/counters.txt | sort -u > $OUT/sq_names.txt
: > $OUT/mix.txt
DT=${1:-mixed}
for set in "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 --dtype $DT > /tmp/pmc.log 2>&1
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step" >> $OUT/mix.txt
  tail -2 /tmp/pmc.log | grep -i "error\|not" >> $OUT/mix.txt
done
cat $OUT/mix.txt
