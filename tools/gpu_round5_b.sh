set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_ref_tests_gpu.py::test_reference_nvbowtie_equals_own_driver_at_3gbp 2>&1 | tail -40 > gpurun_out/suite.log
timeout 600 python tools/fmmap_debug.py > gpurun_out/fmmap_debug.log 2>&1
W=/tmp/w3g
timeout 1500 python tools/nvbowtie_3gbp.py --keep $W --rerun --two-threads --profile gpurun_out/prof3g --json gpurun_out/nvb3g.json --log gpurun_out/nvb3g.log > gpurun_out/nvb3g.out 2>&1
echo "3g rc $?" >> gpurun_out/nvb3g.out
# the same files with the reference layout alone under nvBowtie's kernels
NVBIO_HIP_COMPAT_LINE_NATIVE=0 timeout 600 oracle/_ref/ref_nvBowtie --file-ref -x $W/genome -U $W/reads.fastq -S $W/ref_ln0.sam > gpurun_out/nvb3g_ln0.log 2>&1
grep -v '^@' $W/ref.sam | md5sum > gpurun_out/sam_md5.txt
grep -v '^@' $W/ref_ln0.sam | md5sum >> gpurun_out/sam_md5.txt
grep -v '^@' $W/own.sam | md5sum >> gpurun_out/sam_md5.txt
cd /tmp && NVBIO_HIP_COMPAT_LINE_NATIVE=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof3g_ln0 -o ref_nvbowtie_3gbp_ln0 -- $GRAFT_REPO_ROOT/oracle/_ref/ref_nvBowtie --file-ref -x $W/genome -U $W/reads.fastq -S $W/prof_ln0.sam > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof3g gpurun_out/prof3g_ln0 -name "*.db" -delete 2>/dev/null
find gpurun_out/prof3g gpurun_out/prof3g_ln0 -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh gpurun_out/* | tail -30
