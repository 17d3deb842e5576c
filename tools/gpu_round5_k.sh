set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/w3g
# is the non-blocking device_free what breaks nvBowtie's third batch?  the blocking form first, then the default again on the same files
NVBIO_HIP_SYNC_FREE=1 timeout 700 python tools/nvbowtie_3gbp.py --keep $W --json gpurun_out/nvb3g_syncfree.json --log gpurun_out/nvb3g_syncfree.log > gpurun_out/nvb3g_syncfree.out 2>&1
grep -A12 '"identical"' gpurun_out/nvb3g_syncfree.json | head -20
timeout 700 python tools/nvbowtie_3gbp.py --keep $W --rerun --json gpurun_out/nvb3g_again.json --log gpurun_out/nvb3g_again.log > gpurun_out/nvb3g_again.out 2>&1
grep -A12 '"identical"' gpurun_out/nvb3g_again.json | head -20
