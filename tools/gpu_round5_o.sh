set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/wmt
timeout 200 python tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep $W --json gpurun_out/nvb_1e8.json --log gpurun_out/nvb_1e8.log > gpurun_out/nvb_1e8.out 2>&1
timeout 300 python tools/nvbowtie_poison_probe.py $W "--batch-size 64" > gpurun_out/poison_small.json 2> gpurun_out/poison_small.err
cat gpurun_out/poison_small.json | head -70
W=/tmp/w3g
timeout 400 python tools/nvbowtie_3gbp.py --keep $W --json gpurun_out/nvb3g_p.json --log gpurun_out/nvb3g_p.log > gpurun_out/nvb3g_p.out 2>&1
grep '"identical"' gpurun_out/nvb3g_p.json
timeout 400 python tools/nvbowtie_poison_probe.py $W "" > gpurun_out/poison_3g.json 2> gpurun_out/poison_3g.err
cat gpurun_out/poison_3g.json | head -70
