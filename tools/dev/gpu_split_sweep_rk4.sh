#!/bin/bash
# developer A/B: k_step<double, RK4> time against the step plan (MYO_STEP_SPLIT), one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for sp in "4,3,2,1" "3,3,2,1,1" "3,2,2,1,1,1" "2,2,2,1,1,1,1" "2,2,1,1,1,1,1,1" "3,2,2,2,1" "4,2,2,1,1" "2,2,2,2,1,1"; do
  echo "== f64 rk4 split $sp"; MYO_STEP_SPLIT=$sp python tools/dev/kab.py --dtype f64 --integrator rk4 --rounds 1 --steps 60 myochallenge_amd/libmyobatch.so | grep "mean"
done
for sp in "4,3,2,1" "3,2,2,1,1,1" "2,2,2,1,1,1,1"; do
  echo "== mixed rk4 split $sp"; MYO_STEP_SPLIT=$sp python tools/dev/kab.py --dtype mixed --integrator rk4 --rounds 1 --steps 60 myochallenge_amd/libmyobatch.so | grep "mean"
done
