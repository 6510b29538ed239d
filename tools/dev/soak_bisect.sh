#!/bin/bash
# whole env steps vs the step plan, fp64 P2: state checksums after N steps (developer tool)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
L=myochallenge_amd/libmyobatch.so
for n in "$@"; do
  a=$(MYO_STEP_SPLIT=0 python tools/dev/kab.py $L --dtype f64 --rounds 1 --steps $n --env p2 2>&1 | grep -v amdgpu.ids | tail -1 | awk '{print $NF}')
  b=$(python tools/dev/kab.py $L --dtype f64 --rounds 1 --steps $n --env p2 2>&1 | grep -v amdgpu.ids | tail -1 | awk '{print $NF}')
  c=$(python tools/dev/kab.py $L --dtype f64 --rounds 1 --steps $n --env p2 2>&1 | grep -v amdgpu.ids | tail -1 | awk '{print $NF}')
  d=$(MYO_NO_WRAP_ORDER=1 python tools/dev/kab.py $L --dtype f64 --rounds 1 --steps $n --env p2 2>&1 | grep -v amdgpu.ids | tail -1 | awk '{print $NF}')
  echo "steps $n: whole $a parts $b parts-again $c parts-no-wrap-order $d"
done
