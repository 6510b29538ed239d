#!/bin/bash
# developer A/B: k_step time against the step plan (MYO_STEP_SPLIT), one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for dt in f64 mixed; do
for sp in "4,3,2,1" "3,3,2,1,1" "3,2,2,1,1,1" "4,2,2,1,1" "3,2,2,2,1" "2,2,2,1,1,1,1" "4,3,2,1" "5,3,2" "3,3,2,2" "2,2,2,2,1,1"; do
  echo "== $dt split $sp"; MYO_STEP_SPLIT=$sp python tools/dev/kab.py --dtype $dt --rounds 1 --steps 120 myochallenge_amd/libmyobatch.so | grep "mean"
done
done
echo "== die f64"; for sp in "4,3,2,1" "3,3,2,1,1" "3,2,2,1,1,1"; do echo "split $sp"; MYO_STEP_SPLIT=$sp python tools/dev/kab.py --dtype f64 --env reorient --rounds 1 --steps 120 myochallenge_amd/libmyobatch.so | grep "mean"; done
