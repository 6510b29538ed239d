#!/bin/bash
# Instruction counts per STAGE of a substep (k_step, 4096 envs): the -DMYO_STAGECOUNT build runs one stage twice per pass (myo_physics.h,
# REP); the difference of the launch's counters to the plain pass is the stage.  Run through gpurun:
#   bash tools/dev/stage_counts.sh [f64|mixed] [euler|rk4] [p1|reorient]      -> gpurun_out/stage_counts.txt
# Build first (here): hipcc ... -DMYO_STAGECOUNT csrc/myobatch.hip -o tools/dev/lib_stagecount.so
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/stage_counts.txt
DT=${1:-f64}; IN=${2:-euler}; EN=${3:-p1}
export PYTHONPATH=$ROOT MYO_ALLOW_STALE_LIB=1
cd /tmp && export TMPDIR=/tmp
NAMES=("kinematics" "com_pos" "tendon stage (all)" "tendon: geom wraps" "crb" "constraint set (limits+friction rows+collision)" "collision" "body velocities" "velocity: RNE + passive" "constraint reference" "actuation" "qacc_smooth solve" "newton: H + factor + solve" "newton: load M + hessian" "newton: M/J products of the direction" "newton: update_constraint" "euler: implicit solve")
: > $OUT
for bit in -1 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  if [ $bit -lt 0 ]; then export MYO_DBG_REPEAT=0; label="(plain)"; else export MYO_DBG_REPEAT=$((1 << bit)); label=${NAMES[$bit]}; fi
  rm -rf /tmp/pmc
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmc -- python3 $ROOT/tools/dev/stage_step.py $ROOT/tools/dev/lib_stagecount.so $DT $IN $EN > /tmp/stage.log 2>&1
  echo "== bit $bit $label | $(grep checksum /tmp/stage.log)" >> $OUT
  python3 $ROOT/tools/dev/pmc_summarise.py /tmp/pmc "k_step<" >> $OUT
done
FS=10; [ "$EN" = reorient ] && FS=5
python3 - $OUT $FS <<'PY'
import re, sys
rows, cur = [], None
for l in open(sys.argv[1]):
    m = re.match(r"== bit (-?\d+) (.*) \| checksum (\w+)", l)
    if m:
        cur = {"bit": int(m.group(1)), "name": m.group(2), "sum": m.group(3)}; rows.append(cur); continue
    m = re.match(r"PMC (\S+) (\S+) ", l)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
base = rows[0]
per = 4096 * float(sys.argv[2])      # env-substeps per launch
print("plain: VALU %.0f SALU %.0f LDS %.0f wave-cycles(x4) %.0f per env-substep; checksum %s" % (base["SQ_INSTS_VALU"] / per, base["SQ_INSTS_SALU"] / per, base["SQ_INSTS_LDS"] / per, base["SQ_WAVE_CYCLES"] / per, base["sum"]))
for r in rows[1:]:
    ok = "" if r["sum"] == base["sum"] else "   (CHECKSUM DIFFERS: not repeatable, ignore)"
    d = {k: (r[k] - base[k]) / per for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES")}
    print("%-52s VALU %6.0f (%4.1f %%)  SALU %6.0f  LDS %5.0f  cycles %6.0f (%4.1f %%)%s" % (r["name"], d["SQ_INSTS_VALU"], 100 * d["SQ_INSTS_VALU"] * per / base["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], d["SQ_INSTS_LDS"], d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAVE_CYCLES"] * per / base["SQ_WAVE_CYCLES"], ok))
PY
