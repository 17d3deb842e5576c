set -x
cd /root/repo
python bench.py > gpurun_out/bench_n1.json.log 2> gpurun_out/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python /root/repo/bench.py > /root/repo/gpurun_out/bench_prof.log 2>&1
cp /tmp/pb/*kernel_stats.csv /root/repo/gpurun_out/bench_kernel_stats.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/pf -o f -- python /root/repo/bench.py --only full --no-cpu > /root/repo/gpurun_out/pmc_full.log 2>&1
python /root/repo/tools/pmc_summary.py --each /tmp/pf/*counter_collection.csv full_gotoh > /root/repo/gpurun_out/pmc_full_gotoh_each.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/pd -o d -- python /root/repo/bench.py --only dp --no-cpu > /root/repo/gpurun_out/pmc_dp.log 2>&1
python /root/repo/tools/pmc_summary.py /tmp/pd/*counter_collection.csv banded_gotoh > /root/repo/gpurun_out/pmc_banded.txt 2>&1
cd /root/repo
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
