set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ref_tests_gpu.py -x -q -m gpu -k "fmmap or runs_end_to_end or sw_benchmark" 2>&1 | tail -15 > gpurun_out/t1.log
timeout 900 python -m pytest tests/test_banded_gpu.py tests/test_full_gotoh_gpu.py tests/test_sw_ed_gpu.py tests/test_traceback_gpu.py tests/test_select_gpu.py tests/test_cxx_driver.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/t2.log
timeout 600 python tools/nvbowtie_3gbp.py --genome 1e8 --reads 300000 --repeats 0.6 --two-threads --json gpurun_out/nvb_small.json --log gpurun_out/nvb_small.log > gpurun_out/nvb_small.out 2>&1
echo "small rc $?" >> gpurun_out/nvb_small.out
df -h /tmp | tail -1 > gpurun_out/df.txt; free -g >> gpurun_out/df.txt; nproc >> gpurun_out/df.txt
timeout 1500 python tools/nvbowtie_3gbp.py --two-threads --profile gpurun_out/prof3g --json gpurun_out/nvb3g.json --log gpurun_out/nvb3g.log > gpurun_out/nvb3g.out 2>&1
echo "3g rc $?" >> gpurun_out/nvb3g.out
tail -5 gpurun_out/t1.log gpurun_out/t2.log gpurun_out/nvb_small.out gpurun_out/nvb3g.out
