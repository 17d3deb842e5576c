#!/bin/bash
# Round-6 evidence, in one GPU call: smoke, the default bench line, rocprofv3 kernel stats of the same command, kernel stats of the own C++ driver on the
# repeat-rich 3 Gbp genome, the GPU suite.  Everything lands in gpurun_out/r06/ (copied to profiles/r06/ by hand).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
( time python bench.py --steps 20 --warmup 5 > $O/bench_n1.json.log 2> $O/bench_n1.err ) 2> $O/bench_n1.time
cp $R/gpurun_out/bench_legs.json $O/bench_legs.json 2>/dev/null
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --steps 20 --warmup 5 --no-repeat-rich > $O/bench_prof.log 2>&1
find /tmp/pb -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po -o own -- python $R/tools/own_driver_3gbp.py --workers 1 --no-stage-clock --reps 1 > $O/own_prof.log 2>&1
find /tmp/po -name "*kernel_stats.csv" -exec cp {} $O/own_driver_3gbp_repeats_kernel_stats_full.csv \;
cd $R
python - <<P
import csv
rows=list(csv.reader(open('$O/own_driver_3gbp_repeats_kernel_stats_full.csv')))
keep=[rows[0]]+[r for r in rows[1:] if 'nvb::' in r[0] or 'rocclr' in r[0] or 'ROCPRIM_400200' in r[0]]
csv.writer(open('$O/own_driver_3gbp_repeats_kernel_stats.csv','w')).writerows(keep)
P
rm -f $O/own_driver_3gbp_repeats_kernel_stats_full.csv
# fresh counters of the headline kernel on this round's code: separate --pmc passes (never combined with a trace), summarised per kernel
: > $O/pmc_dp_counters.txt
for set in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc; ( cd /tmp && timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -o c -- python $R/bench.py --only dp --no-cpu > /tmp/pmc.log 2>&1 )
  echo "== $set (bench.py --only dp --no-cpu)" >> $O/pmc_dp_counters.txt
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f banded_gotoh_score_kernel >> $O/pmc_dp_counters.txt 2>&1 || tail -3 /tmp/pmc.log >> $O/pmc_dp_counters.txt
done
( time timeout 3300 python -m pytest tests/ -q -m gpu --durations=8 > $O/gpu_suite.txt 2>&1 ) 2> $O/gpu_suite.time
cp $R/gpurun_out/nvbowtie_3gbp.json $O/nvbowtie_3gbp_se.json 2>/dev/null; cp $R/gpurun_out/nvbowtie_3gbp_paired.json $O/nvbowtie_3gbp_paired.json 2>/dev/null
tail -3 $O/gpu_suite.txt; cat $O/bench_n1.time | tail -3; tail -c 400 $O/bench_n1.json.log; cat $O/smoke.log | tail -1
