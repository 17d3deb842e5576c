"""Where the reference's sw-benchmark (oracle/_ref/ref_sw_benchmark, unchanged) spends its timed enact() calls: runs it on the test's input under
rocprofv3 --hip-trace --kernel-trace --stats and prints the HIP calls and kernels by total time.  GPU box only."""
import csv, glob, os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="swb_")
rnd = random.Random(11)
n_reads, read_len, ref_len = 20000, 150, 16384
ref = "".join(rnd.choice("ACGT") for _ in range(ref_len))
with open(os.path.join(tmp, "ref.fa"), "w") as f:
    f.write(">chr1 synthetic\n" + "\n".join(ref[i:i + 70] for i in range(0, ref_len, 70)) + "\n")
with open(os.path.join(tmp, "reads.fq"), "w") as f:
    for i in range(n_reads):
        p = rnd.randrange(0, ref_len - read_len); r = list(ref[p:p + read_len])
        for _ in range(4):
            r[rnd.randrange(read_len)] = rnd.choice("ACGT")
        f.write("@read%d\n%s\n+\n%s\n" % (i, "".join(r), "I" * read_len))
exe = os.path.join(ROOT, "oracle", "_ref", "ref_sw_benchmark")
wrap = ["rocprofv3", "--hip-trace", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(tmp, "p"), "-o", "s", "--"] if "--plain" not in sys.argv else []
r = subprocess.run(wrap + [exe, os.path.join(tmp, "reads.fq"), os.path.join(tmp, "ref.fa")], capture_output=True, text=True)
print("\n".join(l for l in (r.stdout + r.stderr).replace("\r", "\n").splitlines() if "GCUPS" in l))
for kind in ("hip_api_stats", "kernel_stats"):
    for path in glob.glob(os.path.join(tmp, "p", "*" + kind + ".csv")):
        print("==", kind)
        for row in list(csv.DictReader(open(path)))[:12]:
            print("%9.2f ms %6s calls  %s" % (float(row["TotalDurationNs"]) / 1e6, row["Calls"], row["Name"][:110]))
