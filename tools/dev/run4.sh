cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python tools/dev/kab.py tools/dev/lib_base.so tools/dev/lib_nc.so --rounds 2 2>&1 | tail -4
python tools/dev/kab.py tools/dev/lib_base.so tools/dev/lib_nc.so --rounds 1 --env reorient 2>&1 | tail -3
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
