# round-5 final evidence, second take (selection rebuilds the heap on keys in LDS): selection tests, python bench.py, the same command under
# rocprofv3 --kernel-trace --stats (without the reference-application leg), the 3 Gbp comparison
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_select_gpu.py tests/test_golden_vectors.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/u_select.out 2>&1
tail -3 gpurun_out/u_select.out
timeout 330 python bench.py > gpurun_out/bench_r05.json.log 2> gpurun_out/bench_r05.err
tail -c 300 gpurun_out/bench_r05.json.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --no-ref-app > $R/gpurun_out/bench_prof_r05.log 2>&1
cp /tmp/pb/*kernel_stats.csv $R/gpurun_out/bench_kernel_stats_r05.csv
