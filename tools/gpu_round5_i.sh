set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# a small genome with SA ranges beyond 2^20 rows: 97 % of it one two-symbol period
W=/tmp/wp
timeout 300 python tools/nvbowtie_3gbp.py --genome 3.4e6 --reads 200000 --repeats 0.97 --families 2:1.0:0.00002 --keep $W --json gpurun_out/nvb_periodic.json --log gpurun_out/nvb_periodic.log > gpurun_out/nvb_periodic.out 2>&1
PROBE_ALL_READS=1 timeout 300 python tools/nvbowtie_trace_probe.py $W 0 1 2 > gpurun_out/trace_periodic.json 2> gpurun_out/trace_periodic.err
W=/tmp/w3g
timeout 600 python tools/nvbowtie_3gbp.py --keep $W --json gpurun_out/nvb3g.json --log gpurun_out/nvb3g.log > gpurun_out/nvb3g.out 2>&1
timeout 600 python tools/nvbowtie_trace_probe.py $W 1 3 > gpurun_out/trace_probe.json 2> gpurun_out/trace_probe.err
