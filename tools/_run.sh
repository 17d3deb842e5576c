export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_select_gpu.py tests/test_select_oracle.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --only e2e --e2e-batches 0 --pairs 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())['e2e_leg']
for k in ('cxx_best_approx','cxx_best_approx_hbm_rich'):
    print(k, d[k]['ms_per_batch'], d[k]['stage_ms'], d[k]['identical_to_python_driver'])
print(d['parity'])
"
