#!/bin/bash
# the same run repeated: state checksums must agree (developer tool).  args: lib steps repeats [kab args...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
L=$1; N=$2; R=$3; shift 3
for r in $(seq 1 $R); do
  python tools/dev/kab.py $L --rounds 1 --steps $N "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | awk '{print $NF}'
done | sort | uniq -c | sed "s|^|$L $N $* : |"
