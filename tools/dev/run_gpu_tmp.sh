export PYTHONPATH=$PWD
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python tools/dev/gpu_rec_dbg.py 4096 2>&1 | grep "train\|collect" | cut -c1-120
timeout 300 python tools/bench_reorient.py 2>&1 | tail -1
