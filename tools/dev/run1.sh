cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q -k "rk4_mixed or config_c or whole_episode" 2>&1 | tail -15 > gpurun_out/r03/new_tests.log
python tools/dev/kab.py tools/dev/lib_base.so tools/dev/lib_rl.so --rounds 3 > gpurun_out/r03/kab1.log 2>&1
python tools/dev/gpu_prof.py 4096 tools/dev/lib_prof_base.so f32 > gpurun_out/r03/prof_base.log 2>&1
python tools/dev/gpu_prof.py 4096 tools/dev/lib_prof_rl.so f32 > gpurun_out/r03/prof_rl.log 2>&1
cat gpurun_out/r03/new_tests.log gpurun_out/r03/kab1.log
