set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/w3g
timeout 600 python tools/nvbowtie_3gbp.py --keep $W --json gpurun_out/nvb3g.json --log gpurun_out/nvb3g.log > gpurun_out/nvb3g.out 2>&1
timeout 600 python tools/nvbowtie_trace_probe.py $W 0 1 3 > gpurun_out/trace_probe.json 2> gpurun_out/trace_probe.err
