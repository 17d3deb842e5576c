export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/test_select_gpu.py tests/test_select_oracle.py tests/test_pipeline_gpu.py tests/test_all_mapping_gpu.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --only e2e --e2e-batches 0 --pairs 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())['e2e_leg']
for k in ('cxx_best_approx_hbm_rich',):
    print(k, d[k]['ms_per_batch'], d[k]['stage_ms'], d[k]['identical_to_python_driver'])
"
timeout 900 python tools/own_driver_3gbp.py --check --workers 1,2 --json gpurun_out/r06/own_driver_3gbp_select_regs.json > /dev/null 2>&1
python - <<P
import json
d=json.load(open('gpurun_out/r06/own_driver_3gbp_select_regs.json'))
print(d['one_batch']['ms_per_batch'], d['one_batch']['stage_ms_with_syncs'], d.get('python_driver'), {k:v['Mreads_per_s'] for k,v in d['pipelined'].items()})
P
