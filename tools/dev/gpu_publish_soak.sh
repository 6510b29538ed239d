#!/bin/bash
# long soak of the write-through hand-off: state checksum of whole env steps == the step plan with write-through records,
# 20,000 env steps x 4096 P2 envs (with resets) per stepper, 6,000 for the die
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/publish_soak; mkdir -p $O
L=myochallenge_amd/libmyobatch.so
for dt in f64 mixed; do
  MYO_STEP_SPLIT=0 python tools/dev/kab.py $L --dtype $dt --rounds 1 --steps 20000 --env p2 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/whole $dt p2: /"
  python tools/dev/kab.py $L --dtype $dt --rounds 1 --steps 20000 --env p2 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/parts $dt p2: /"
done
MYO_STEP_SPLIT=0 python tools/dev/kab.py $L --dtype f64 --rounds 1 --steps 6000 --env reorient 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/whole f64 die: /"
python tools/dev/kab.py $L --dtype f64 --rounds 1 --steps 6000 --env reorient 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/parts f64 die: /"
