# rocprofv3 counter passes for the FM-index kernels on this round's code (profiles/traffic.json: fm_rank / fm_match / fm_locate and their
# line-native forms).  Counter-only runs (no --kernel-trace / --stats with --pmc).  GPU box; writes gpurun_out/pmc_fm_*.txt
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for leg in rank seed; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d /tmp/pfm_rd_$leg -o f -- python $R/bench.py --only $leg --no-cpu > $R/gpurun_out/pmc_fm_rd_$leg.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pfm_rd_$leg/*counter_collection.csv fm_ > $R/gpurun_out/pmc_fm_rd_$leg.txt 2>&1
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d /tmp/pfm_wr_$leg -o f -- python $R/bench.py --only $leg --no-cpu > $R/gpurun_out/pmc_fm_wr_$leg.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pfm_wr_$leg/*counter_collection.csv fm_ > $R/gpurun_out/pmc_fm_wr_$leg.txt 2>&1
done
