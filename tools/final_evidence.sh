# round-end evidence on the GPU box: python bench.py (N = 1), the same command under rocprofv3 --kernel-trace --stats, counter passes for
# the two DP kernels, smoke(), the reference's nvBowtie at speed and its kernel statistics.  Writes under gpurun_out/; copy what is judged
# into profiles/rNN/.
set -x
cd /root/repo
python bench.py > gpurun_out/bench_n1.json.log 2> gpurun_out/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python /root/repo/bench.py > /root/repo/gpurun_out/bench_prof.log 2>&1
cp /tmp/pb/*kernel_stats.csv /root/repo/gpurun_out/bench_kernel_stats.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/pf -o f -- python /root/repo/bench.py --only full --no-cpu > /root/repo/gpurun_out/pmc_full.log 2>&1
python /root/repo/tools/pmc_summary.py --each /tmp/pf/*counter_collection.csv full_gotoh > /root/repo/gpurun_out/pmc_full_gotoh_each.txt 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS --output-format csv -d /tmp/pf2 -o f -- python /root/repo/bench.py --only full --no-cpu > /root/repo/gpurun_out/pmc_full2.log 2>&1
python /root/repo/tools/pmc_summary.py --each /tmp/pf2/*counter_collection.csv full_gotoh > /root/repo/gpurun_out/pmc_full_gotoh_l2.txt 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/pf3 -o f -- python /root/repo/bench.py --only full --no-cpu > /root/repo/gpurun_out/pmc_full3.log 2>&1
python /root/repo/tools/pmc_summary.py --each /tmp/pf3/*counter_collection.csv full_gotoh > /root/repo/gpurun_out/pmc_full_gotoh_lds.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/pd -o d -- python /root/repo/bench.py --only dp --no-cpu > /root/repo/gpurun_out/pmc_dp.log 2>&1
python /root/repo/tools/pmc_summary.py /tmp/pd/*counter_collection.csv banded_gotoh > /root/repo/gpurun_out/pmc_banded.txt 2>&1
python /root/repo/tools/nvbowtie_speed.py --reads 4000000 > /root/repo/gpurun_out/nvbowtie_speed4.log 2>&1
python /root/repo/tools/nvbowtie_speed.py --reads 1000000 --wrap "rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -o n --" > /root/repo/gpurun_out/nvbowtie_prof1.log 2>&1
cp /tmp/pn/*kernel_stats.csv /root/repo/gpurun_out/nvbowtie_kernel_stats.csv
cd /root/repo
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
