set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_select_gpu.py -q -m gpu -x -k "wide" 2>&1 | tail -25 > gpurun_out/t5.log
