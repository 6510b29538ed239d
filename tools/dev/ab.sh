#!/bin/bash
# A/B timing of prebuilt library variants tools/dev/lib_<name>.so on ONE box: bash tools/dev/ab.sh name1 name2 ...
cp myochallenge_amd/libmyobatch.so /tmp/lib_keep.so
for r in 1 2; do for v in "$@"; do
  cp tools/dev/lib_$v.so myochallenge_amd/libmyobatch.so
  python bench.py --no-cpu-baseline --no-variants --min-seconds 1.5 > /tmp/b_$v.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.loads(open("/tmp/b_$v.json").read().strip().splitlines()[-1]); print("$v", round(d["value"]), round(d["ms_per_step"],4), round(d["env_kernel_ms"],4))
except Exception as e: print("$v", "FAILED", e)
PY
done; done
cp /tmp/lib_keep.so myochallenge_amd/libmyobatch.so
