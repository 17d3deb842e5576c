set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_compat_alignment_gpu.py tests/test_compat_filter.py tests/test_compat_fmindex.py tests/test_bench_multirank_gpu.py tests/test_io_formats.py -q -m gpu 2>&1 | tail -25 > gpurun_out/t3.log
timeout 600 python -m pytest tests/test_ref_tests_gpu.py -q -m gpu -k "fmmap" 2>&1 | tail -8 >> gpurun_out/t3.log
W=/tmp/w3g
timeout 1500 python tools/nvbowtie_3gbp.py --keep $W --profile gpurun_out/prof3g --json gpurun_out/nvb3g.json --log gpurun_out/nvb3g.log > gpurun_out/nvb3g.out 2>&1
echo "3g rc $?" >> gpurun_out/nvb3g.out
timeout 900 python tools/nvbowtie_mt_probe.py $W > gpurun_out/mt_probe.json 2> gpurun_out/mt_probe.err
# without multi-hit rounds: the per-read differences alone (no hits-per-read cascade)
cp $W/ref.sam $W/ref_multi.sam; cp $W/own.sam $W/own_multi.sam
timeout 900 python - <<'PY' > gpurun_out/nvb3g_nomulti.json 2> gpurun_out/nvb3g_nomulti.err
import sys, os, json, subprocess, time
sys.path.insert(0, "tools")
import numpy as np, torch
import nvbowtie_3gbp as T
W = "/tmp/w3g"
exe = "oracle/_ref/ref_nvBowtie"
r = subprocess.run([exe, "--no-multi-hits", "1", "--file-ref", "-x", W + "/genome", "-U", W + "/reads.fastq", "-S", W + "/ref_nm.sam"], capture_output=True, text=True)
# the reads back from the FASTQ file (fixed-size records)
raw = np.fromfile(W + "/reads.fastq", dtype=np.uint8).reshape(-1, 215)
lut = np.full(256, 4, np.uint8)
for c, v in zip(b"ACGT", range(4)): lut[c] = v
sym = torch.from_numpy(lut[raw[:, 11:111]]).cuda(); qual = torch.from_numpy(raw[:, 113:213] - 33).cuda()
out = {}
T.own_driver(W + "/genome", sym, qual, W + "/own_nm.sam", torch.device("cuda:0"), 1 << 20, timings=out, overrides=dict(no_multi_hits=True))
n_a, n_b, same, diffs, cats = T.compare_sam(W + "/ref_nm.sam", W + "/own_nm.sam", show=12)
out.update(exit=r.returncode, records_ref=n_a, records_own=n_b, identical=same, categories=cats, first_differences=diffs)
print(json.dumps(out, indent=1, default=str))
PY
find gpurun_out/prof3g -name "*kernel_trace.csv" -delete 2>/dev/null
bash tools/pmc_fm_refresh.sh
du -sh gpurun_out/prof3g
