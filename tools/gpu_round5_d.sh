set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/w3g
timeout 600 python tools/nvbowtie_3gbp.py --keep $W --json gpurun_out/nvb3g.json --log gpurun_out/nvb3g.log > gpurun_out/nvb3g.out 2>&1
echo "3g rc $?" >> gpurun_out/nvb3g.out
timeout 600 python tools/nvbowtie_nomulti_probe.py > gpurun_out/nvb3g_nomulti.json 2> gpurun_out/nvb3g_nomulti.err
timeout 500 python tools/nvbowtie_mt_probe.py $W > gpurun_out/mt_probe.json 2> gpurun_out/mt_probe.err
timeout 500 python tools/nvbowtie_mt_probe.py $W NVBIO_HIP_SYNC_FREE=1 > gpurun_out/mt_probe_syncfree.json 2> gpurun_out/mt_probe_syncfree.err
timeout 300 python -m pytest tests/test_compat_alignment_gpu.py -q -m gpu -k bitvector 2>&1 | tail -5 > gpurun_out/t4.log
