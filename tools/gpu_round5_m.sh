set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# the whole GPU suite on the round's final code (includes the 3 Gbp own-vs-nvBowtie test), then the two-thread mode of nvBowtie under launch blocking
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/m_gpu_suite.out 2>&1
tail -5 gpurun_out/m_gpu_suite.out
W=/tmp/wmt
timeout 200 python tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep $W --json gpurun_out/nvb_1e8.json --log gpurun_out/nvb_1e8.log > gpurun_out/nvb_1e8.out 2>&1
grep '"identical"' gpurun_out/nvb_1e8.json
MT_MATRIX_QUICK=1 timeout 400 python tools/nvbowtie_mt_matrix.py $W > gpurun_out/mt_matrix_r05b.json 2> gpurun_out/mt_matrix_r05b.err
cat gpurun_out/mt_matrix_r05b.json
