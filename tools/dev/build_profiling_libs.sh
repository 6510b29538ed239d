#!/bin/bash
# the two instrumented builds tools/profile_round.sh runs (stage shares, workgroup timeline); run here before gpurun
set -e
cd "$(dirname "$0")/../.."
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fno-hip-fp32-correctly-rounded-divide-sqrt"
/opt/rocm/bin/hipcc $F -DMYO_PROF myochallenge_amd/csrc/myobatch.hip -o tools/dev/libmyobatch_prof.so
/opt/rocm/bin/hipcc $F -DMYO_WGTIME myochallenge_amd/csrc/myobatch.hip -o tools/dev/lib_wgtime.so
