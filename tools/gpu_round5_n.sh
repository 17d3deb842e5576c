set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/wmt
timeout 200 python tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep $W --json gpurun_out/nvb_1e8.json --log gpurun_out/nvb_1e8.log > gpurun_out/nvb_1e8.out 2>&1
timeout 400 python tools/nvbowtie_batch_order_probe.py $W 64 > gpurun_out/batch_order.json 2> gpurun_out/batch_order.err
cat gpurun_out/batch_order.json | head -80
