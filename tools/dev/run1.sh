cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q -k "rk4_mixed or config_c or whole_episode" 2>&1 | tail -15 > gpurun_out/r03/new_tests.log
python tools/dev/gpu_prof.py 4096 tools/dev/lib_prof_base.so f32 > gpurun_out/r03/prof_base.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --help 2>&1 | grep -i -A3 "att" | head -40 > $GRAFT_REPO_ROOT/gpurun_out/r03/att_help.log
ls /opt/rocm/lib | grep -i "trace\|decoder" >> $GRAFT_REPO_ROOT/gpurun_out/r03/att_help.log
cd $GRAFT_REPO_ROOT
cat gpurun_out/r03/new_tests.log gpurun_out/r03/prof_base.log
