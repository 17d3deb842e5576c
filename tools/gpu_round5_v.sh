# the whole GPU suite on the round's final build
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 340 python -m pytest tests -m gpu -x -q > gpurun_out/v_gpu_suite.out 2>&1
tail -4 gpurun_out/v_gpu_suite.out
