#!/bin/bash
# Regenerates the measured records of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh
# Outputs land in gpurun_out/round/ ; copy the ones to keep into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/round
mkdir -p $OUT
export PYTHONPATH=$ROOT
python3 -c "from myochallenge_amd import native; print(native.load().build_id)" > $OUT/build_id.txt
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (with the CPU baseline leg)
python3 $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench.err      # includes the f64 / RK4 variants and the CPU leg
# 2. per-kernel time of the same command (no CPU leg: it is host-only work)
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants > /tmp/stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
grep -v "^[WIE]2026" /tmp/stats.log | tail -1 > $OUT/bench_line_under_rocprof.json
# 3. PMC counters of k_step inside the same command, one --pmc pass per set (no other tracing)
: > $OUT/pmc_kstep.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LDS_BANK_CONFLICT" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 > /tmp/pmc.log 2>&1
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT/pmc_kstep.txt
done
# 3a. HBM traffic of the same launches with WHOLE env steps (MYO_STEP_SPLIT=0): what the hand-offs of the step plan add
: > $OUT/pmc_kstep_whole_steps.txt
export MYO_STEP_SPLIT=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 > /tmp/pmc.log 2>&1
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT/pmc_kstep_whole_steps.txt
done
unset MYO_STEP_SPLIT
# 3a'. ... and with the parts published by an agent release fence (round 4's form) instead of write-through stores
: > $OUT/pmc_kstep_publish_fence.txt
export MYO_PUBLISH=fence
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 > /tmp/pmc.log 2>&1
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT/pmc_kstep_publish_fence.txt
done
python3 $ROOT/bench.py --no-cpu-baseline --no-variants > $OUT/bench_publish_fence.json 2>/dev/null
unset MYO_PUBLISH
# 3a''. dynamic instruction classes of k_step<double> (fp64 / fp32 / integer / the rest = moves, selects, compares,
#       cross-lane; LDS / scalar / branch counts) — one pass per set
: > $OUT/pmc_instruction_mix_f64.txt
for set in "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH_LEVEL"; do
  rm -rf /tmp/pmc
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 > /tmp/pmc.log 2>&1
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT/pmc_instruction_mix_f64.txt
done
# 3b. matrix-core activity of the PPO side (north_star asks for MFMA-busy against peak): hipBLASLt GEMM kernels
rm -rf /tmp/pmc
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 > /tmp/pmc.log 2>&1
python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "Cijk" > $OUT/pmc_gemm.txt
python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT/pmc_gemm.txt
# 3c. matrix cores in the fp64 stepper (block-arrow Newton solve: Schur complement + 16 x 16 factor)
rm -rf /tmp/pmc
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 --dtype f64 > /tmp/pmc.log 2>&1
python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" > $OUT/pmc_mfma_kstep_f64.txt
# 4. stage shares (instrumented build; shares only)
PYTHONPATH=$ROOT python3 $ROOT/tools/dev/gpu_prof.py > $OUT/stage_shares.txt 2>&1
PYTHONPATH=$ROOT python3 $ROOT/tools/dev/gpu_prof.py 4096 $ROOT/tools/dev/libmyobatch_prof.so f32 rk4 > $OUT/stage_shares_rk4.txt 2>&1
# 4b. workgroup timeline of one k_step launch: whole steps against the step plan (slots busy, makespan)
MYO_STEP_SPLIT=0 PYTHONPATH=$ROOT python3 $ROOT/tools/dev/gpu_wgtime.py $ROOT/tools/dev/lib_wgtime.so 4096 1 > $OUT/wg_timeline_whole.log 2>&1
PYTHONPATH=$ROOT python3 $ROOT/tools/dev/gpu_wgtime.py $ROOT/tools/dev/lib_wgtime.so 4096 4 > $OUT/wg_timeline_parts.log 2>&1
# 5. other configurations (BASELINE.json configs / variants)
python3 $ROOT/bench.py --no-cpu-baseline --no-variants --dtype mixed > $OUT/bench_mixed.json 2>/dev/null
python3 $ROOT/bench.py --no-cpu-baseline --no-variants --integrator rk4 > $OUT/bench_rk4.json 2>/dev/null
python3 $ROOT/bench.py --no-cpu-baseline --no-variants --env-name CustomMyoBaodingBallsP2 --envs 8192 > $OUT/bench_p2_8192.json 2>/dev/null
python3 $ROOT/bench.py --no-cpu-baseline --no-variants --no-ppo > $OUT/bench_rollout_only.json 2>/dev/null
python3 $ROOT/bench.py --no-cpu-baseline --no-variants --env-name CustomMyoReorientP2 > $OUT/bench_reorient_p2.json 2>/dev/null
python3 $ROOT/tools/bench_reorient.py > $OUT/bench_reorient_lstm.json 2>/dev/null      # the light setting (32-step rollouts, 4 epochs), fp64 since round 6
python3 $ROOT/tools/bench_reorient.py --reference-settings --iters 6 > $OUT/bench_reorient_lstm_reference.json 2>/dev/null   # n_steps 128, n_epochs 10, fp64 (src/main_reorient.py:53-71)
MYO_LSTM_SEQ=0 python3 $ROOT/tools/bench_reorient.py --reference-settings --iters 4 > $OUT/bench_reorient_lstm_reference_step_kernels.json 2>/dev/null   # the same with one launch per LSTM time step
rm -rf /tmp/prof_cfge
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfge -o e -- python3 $ROOT/tools/bench_reorient.py --reference-settings --iters 2 > /tmp/cfge.log 2>&1
find /tmp/prof_cfge -name "*kernel_stats.csv" -exec cp {} $OUT/config_e_kernel_stats.csv \;
{ RS=4 python3 $ROOT/tools/dev/gpu_lstm_seq_time.py 256 512 128; RS=2 python3 $ROOT/tools/dev/gpu_lstm_seq_time.py 128 512 128; RS=1 python3 $ROOT/tools/dev/gpu_lstm_seq_time.py 256 512 128; } > $OUT/lstm_seq_time.txt 2>&1
python3 $ROOT/bench.py --no-cpu-baseline --no-variants --dtype mixed --lstm-hidden 128 --net-arch "" --n-steps 32 > $OUT/bench_lstm128.json 2>/dev/null   # the reference's phase-1 policy shape
# 6. trajectory drift tables of both steppers (32 action streams x 200 env steps) and k_step time against the batch size
python3 $ROOT/tools/dev/gpu_drift.py > $OUT/drift.log 2>&1
# 6b. training demos (deterministic evaluation before / during PPO training; prints the batch health counters at the end)
python3 $ROOT/tools/train_demo.py --steps 100000000 --dtype f64 --out $OUT/train_demo_p1_100m_f64.json > $OUT/train_demo_p1.log 2>&1
python3 $ROOT/tools/train_demo.py --env-name CustomMyoReorientP1 --steps 40000000 --lstm-hidden 256 --out $OUT/train_demo_reorient_lstm256_40m.json > $OUT/train_demo_reorient.log 2>&1
# 7. the GPU test suite on the same library
cd $ROOT && python3 -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
echo done
