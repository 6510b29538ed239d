#!/bin/bash
# Where k_step's memory-side write bytes come from: WRITE_SIZE / FETCH_SIZE of the bench under the step plan's A/B switches.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmcw
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc.txt
run() {   # label, counter, "bench args", env assignments...
  local label=$1 ctr=$2 bargs=$3; shift 3
  rm -rf /tmp/pmc
  ( for kv in "$@"; do export "$kv"; done
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 $bargs > /tmp/pmc.log 2>&1 )
  echo "== $label" >> $OUT/pmc.txt
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT/pmc.txt
}
run "f64 default plan" WRITE_SIZE ""
run "f64 whole steps" WRITE_SIZE "" MYO_STEP_SPLIT=10
run "f64 whole steps, 2048 envs" WRITE_SIZE "--envs 2048" MYO_STEP_SPLIT=10
run "mixed whole steps" WRITE_SIZE "--dtype mixed" MYO_STEP_SPLIT=10
run "mixed default plan" WRITE_SIZE "--dtype mixed"
cat $OUT/pmc.txt
