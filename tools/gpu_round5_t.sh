# round-5 final evidence on the GPU box: python bench.py (N = 1), the same command under rocprofv3 --kernel-trace --stats, one counter pass over
# the FM-index legs (TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum, per dispatch), smoke().  Writes under gpurun_out/.
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 330 python bench.py > gpurun_out/bench_r05.json.log 2> gpurun_out/bench_r05.err
tail -c 600 gpurun_out/bench_r05.json.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r05.log 2>&1
cd /tmp
timeout 330 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --no-ref-app > $R/gpurun_out/bench_prof_r05.log 2>&1
cp /tmp/pb/*kernel_stats.csv $R/gpurun_out/bench_kernel_stats_r05.csv
for leg in rank seed; do
  timeout 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/pfm_$leg -o f -- python $R/bench.py --only $leg --no-cpu > $R/gpurun_out/pmc_fm_$leg.log 2>&1
  python $R/tools/pmc_summary.py --each /tmp/pfm_$leg/*counter_collection.csv fm_ > $R/gpurun_out/pmc_fm_each_$leg.txt 2>&1
done
