set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
W=/tmp/wmt
timeout 200 python tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep $W --json gpurun_out/nvb_1e8.json --log gpurun_out/nvb_1e8.log > gpurun_out/nvb_1e8.out 2>&1
MT_EXAMPLES_CASES=nopool timeout 600 python tools/nvbowtie_mt_examples.py $W > gpurun_out/mt_examples_nopool.json 2> gpurun_out/mt_examples_nopool.err
cut -c1-200 gpurun_out/mt_examples_nopool.err
python - <<EOF
import json
d=json.load(open("gpurun_out/mt_examples_nopool.json"))
for k,v in d.items():
    print(k, v if not isinstance(v,dict) else (v["differ"], v["per_batch"], v["offset_in_batch_min_max"], v["per_256_reads"]))
EOF
