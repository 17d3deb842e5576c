#!/usr/bin/env python3
"""What do the records look like that nvBowtie's two-thread mode (`--device 0 --device 0`) changes against its single-thread run?  Small genome, small
batches (files of a `tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep DIR` run); per case: how many records differ, in which batches, of which
kind (position / score / CIGAR only / MAPQ only), a few examples.  GPU box only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def records(path):
    out = {}
    for l in open(path, "rb").read().split(b"\n"):
        if l and not l.startswith(b"@"):
            out[l.split(b"\t", 1)[0]] = l.split(b"\t")
    return out


def main():
    W = sys.argv[1]
    batch_k = 64
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    base_cmd = ["--batch-size", str(batch_k), "--file-ref", "-x", os.path.join(W, "genome"), "-U", os.path.join(W, "reads.fastq")]

    def run(tag, mt, env):
        sam = os.path.join(W, tag + ".sam")
        try:
            r = subprocess.run([exe] + (["--device", "0", "--device", "0"] if mt else []) + base_cmd + ["-S", sam], capture_output=True, text=True, timeout=60, env=dict(os.environ, **env))
        except subprocess.TimeoutExpired:
            return None
        return records(sam) if r.returncode == 0 else None

    st = run("st", False, {})
    out = {}
    gen = {"NVBIO_HIP_COMPAT_GENERIC": "banded,full,traceback"}
    pool = {"NVBIO_HIP_ROCM_POOL": "1"}                    # the hipMemPool of rounds 1-4 (csrc/banded_gotoh.hip): what every case but "final" was about
    which = os.environ.get("MT_EXAMPLES_CASES", "final")
    if which == "two_processes":
        # two single-thread processes side by side on the one GPU: does a neighbour on the device (not in the process) do it?
        import threading
        for rep in range(3):
            res = [None, None]
            def go(j):
                res[j] = run("proc%d_%d" % (rep, j), False, pool)
            th = [threading.Thread(target=go, args=(j,)) for j in range(2)]
            [t.start() for t in th]; [t.join() for t in th]
            out["two_processes_%d" % rep] = [None if r is None else sum(1 for k, a in st.items() if r.get(k) != a) for r in res]
        cases = []
    elif which == "serialized":
        ser = dict(pool, AMD_SERIALIZE_KERNEL="3", AMD_SERIALIZE_COPY="3")
        cases = [("pool_generic_serialized_%d" % k, dict(gen, **ser)) for k in range(2)] + [("pool_generic_one_hw_queue_%d" % k, dict(gen, GPU_MAX_HW_QUEUES="1", **pool)) for k in range(2)] + \
                [("pool_generic_blocking_free_%d" % k, dict(gen, NVBIO_HIP_SYNC_FREE="1", **pool)) for k in range(2)] + [("pool_one_omp_thread_%d" % k, dict(pool, OMP_NUM_THREADS="1")) for k in range(2)]
    else:
        cases = [("two_threads_generic_%d" % k, gen) for k in range(4)] + [("two_threads_%d" % k, {}) for k in range(4)] + \
                [("two_threads_generic_rocm_pool_%d" % k, dict(gen, **pool)) for k in range(3)]
    for tag, env in cases:
        mt = run(tag, True, env)
        if mt is None or st is None:
            out[tag] = "failed or hung"; continue
        kinds, per_batch, ex = {}, {}, []
        for k, a in st.items():
            b = mt.get(k)
            if b == a:
                continue
            if b is None:
                kind = "missing"
            elif a[3] != b[3] or a[2] != b[2] or (int(a[1]) & 20) != (int(b[1]) & 20):
                kind = "placement"
            elif len(a) > 12 and len(b) > 12 and a[12] != b[12]:
                kind = "score_at_same_placement"
            elif a[5] != b[5]:
                kind = "cigar_only"
            elif a[4] != b[4]:
                kind = "mapq_only"
            else:
                kind = "other"
            kinds[kind] = kinds.get(kind, 0) + 1
            bt = int(k[1:]) // (batch_k * 1024)
            per_batch[bt] = per_batch.get(bt, 0) + 1
            if len(ex) < 4 and b is not None:
                ex.append([[f.decode() for f in a[:9] + a[11:14]], [f.decode() for f in b[:9] + b[11:14]]])
        offs = sorted(int(k[1:]) % (batch_k * 1024) for k, a in st.items() if mt.get(k) != a)
        hist = {}
        for o in offs:
            hist[o >> 8] = hist.get(o >> 8, 0) + 1
        out[tag] = dict(differ=sum(kinds.values()), kinds=kinds, per_batch=per_batch, examples=ex, offset_in_batch_min_max=[offs[0], offs[-1]] if offs else None, per_256_reads=hist)
        print(tag, json.dumps(out[tag])[:300], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
