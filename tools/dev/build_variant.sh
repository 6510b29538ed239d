#!/bin/bash
# bash tools/dev/build_variant.sh <name> [extra hipcc flags]  ->  tools/dev/lib_<name>.so (A/B candidates for tools/dev/kab.py)
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-hip-fp32-correctly-rounded-divide-sqrt "$@" \
  myochallenge_amd/csrc/myobatch.hip -o tools/dev/lib_$name.so
