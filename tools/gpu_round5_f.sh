set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/w3g
timeout 600 python tools/nvbowtie_3gbp.py --keep $W --two-threads --json gpurun_out/nvb3g.json --log gpurun_out/nvb3g.log > gpurun_out/nvb3g.out 2>&1
echo "3g rc $?" >> gpurun_out/nvb3g.out
timeout 900 python tools/nvbowtie_subset_probe.py $W > gpurun_out/subset_probe.json 2> gpurun_out/subset_probe.err
