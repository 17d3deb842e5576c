set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# the rebuilt hit deque (make_interval_heap on every hits[read_id]): device replay + selection rounds vs the oracle, then own driver vs nvBowtie
timeout 400 python -m pytest tests/test_select_gpu.py tests/test_golden_vectors.py -m gpu -x -q > gpurun_out/j_select.out 2>&1
tail -3 gpurun_out/j_select.out
timeout 300 python tools/nvbowtie_3gbp.py --genome 3.4e6 --reads 200000 --repeats 0.97 --families 2:1.0:0.00002 --json gpurun_out/nvb_periodic.json --log gpurun_out/nvb_periodic.log > gpurun_out/nvb_periodic.out 2>&1
tail -5 gpurun_out/nvb_periodic.out
timeout 700 python tools/nvbowtie_3gbp.py --json gpurun_out/nvb3g.json --log gpurun_out/nvb3g.log > gpurun_out/nvb3g.out 2>&1
tail -12 gpurun_out/nvb3g.out
