#!/bin/bash
# developer A/B: k_step<double> time against the step plan (MYO_STEP_SPLIT), one box
for sp in "4,3,2,1" "5,3,2" "6,3,1" "5,4,1" "7,3" "6,4" "4,3,3" "3,3,2,2" "0"; do
  echo "== split $sp"; MYO_STEP_SPLIT=$sp python tools/dev/kab.py --dtype f64 --rounds 1 myochallenge_amd/libmyobatch.so | grep "round 0"
done
