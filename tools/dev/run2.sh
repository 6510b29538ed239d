cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_mlp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mlp -o m -- python3 $GRAFT_REPO_ROOT/tools/dev/gpu_mlp_check.py > /tmp/mlp.log 2>&1
find /tmp/prof_mlp -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r03/mlp_kernel_stats.csv \;
cd $GRAFT_REPO_ROOT
python tools/dev/stats_top.py gpurun_out/r03/mlp_kernel_stats.csv | head -30
