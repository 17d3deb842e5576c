set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/wmt
timeout 200 python tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep $W --json gpurun_out/nvb_1e8.json --log gpurun_out/nvb_1e8.log > gpurun_out/nvb_1e8.out 2>&1
timeout 500 python tools/nvbowtie_mt_examples.py $W > gpurun_out/mt_examples.json 2> gpurun_out/mt_examples.err
cat gpurun_out/mt_examples.err
