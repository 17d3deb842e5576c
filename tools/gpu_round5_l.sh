set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/w3g
NVBIO_HIP_FREE_MODE=1 timeout 700 python tools/nvbowtie_3gbp.py --keep $W --json gpurun_out/nvb3g_mode1.json --log gpurun_out/nvb3g_mode1.log > gpurun_out/nvb3g_mode1.out 2>&1
grep '"identical"' gpurun_out/nvb3g_mode1.json
timeout 600 python tools/nvbowtie_free_modes.py $W 1 0 2 3 4 0 > gpurun_out/free_modes.json 2> gpurun_out/free_modes.err
cat gpurun_out/free_modes.json | head -100
