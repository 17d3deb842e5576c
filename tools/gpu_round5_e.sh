set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/wsmall
timeout 300 python tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep $W --json gpurun_out/nvb_small.json --log gpurun_out/nvb_small.log > gpurun_out/nvb_small.out 2>&1
timeout 900 python tools/nvbowtie_mt_matrix.py $W > gpurun_out/mt_matrix.json 2> gpurun_out/mt_matrix.err
