set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# the block cache in place of the hipMemPool: nvBowtie's two-thread mode, small and at 3 Gbp, then the whole GPU suite
W=/tmp/wmt
timeout 200 python tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep $W --json gpurun_out/nvb_1e8.json --log gpurun_out/nvb_1e8.log > gpurun_out/nvb_1e8.out 2>&1
timeout 600 python tools/nvbowtie_mt_examples.py $W > gpurun_out/two_threads_final.json 2> gpurun_out/two_threads_final.err
cut -c1-160 gpurun_out/two_threads_final.err
timeout 900 python tools/nvbowtie_3gbp.py --two-threads --json gpurun_out/nvb3g_two_threads.json --log gpurun_out/nvb3g_two_threads.log > gpurun_out/nvb3g_two_threads.out 2>&1
grep -E '"identical"|two_threads|nvbowtie_wall' gpurun_out/nvb3g_two_threads.json | head
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s_gpu_suite.out 2>&1
tail -5 gpurun_out/s_gpu_suite.out
