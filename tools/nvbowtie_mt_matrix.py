#!/usr/bin/env python3
"""nvBowtie with two compute threads on one device against its single-thread run, on a small genome with small batches (many batches per thread,
seconds per run), under the drop-in layer's switches one at a time -- to find which part of the layer the two-thread mode trips over.
Works on the files a `tools/nvbowtie_3gbp.py --genome 1e8 --reads 1000000 --keep DIR` run left.  GPU box only."""
import json
import os
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def body(path):
    return Counter(l for l in open(path, "rb").read().split(b"\n") if l and not l.startswith(b"@"))


def main():
    W = sys.argv[1]
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    base = ["--batch-size", "64", "--file-ref", "-x", os.path.join(W, "genome"), "-U", os.path.join(W, "reads.fastq")]
    out = {}

    def run(tag, mt, env):
        sam = os.path.join(W, tag + ".sam")
        try:
            r = subprocess.run([exe] + (["--device", "0", "--device", "0"] if mt else []) + base + ["-S", sam], capture_output=True, text=True, timeout=60, env=dict(os.environ, **env))
        except subprocess.TimeoutExpired:
            return "HUNG"
        return body(sam) if r.returncode == 0 else "exit %d" % r.returncode

    st = run("st", False, {})
    st2 = run("st2", False, {})
    out["single_thread_repeatable"] = (st == st2) if not isinstance(st, str) else st
    quick = [("default", {}), ("default_again", {}), ("launch_blocking", {"HIP_LAUNCH_BLOCKING": "1"}), ("launch_blocking_again", {"HIP_LAUNCH_BLOCKING": "1"}),
             ("serialized", {"AMD_SERIALIZE_KERNEL": "3", "AMD_SERIALIZE_COPY": "3"}), ("sync_free", {"NVBIO_HIP_SYNC_FREE": "1"})]
    cases = [("default", {}), ("default_again", {}), ("sync_free", {"NVBIO_HIP_SYNC_FREE": "1"}), ("no_line_native", {"NVBIO_HIP_COMPAT_LINE_NATIVE": "0"}),
             ("generic_lanes", {"NVBIO_HIP_COMPAT_GENERIC": "banded,full,traceback"}), ("no_views", {"NVBIO_HIP_COMPAT_NO_VIEWS": "1"}),
             ("sync_free+generic", {"NVBIO_HIP_SYNC_FREE": "1", "NVBIO_HIP_COMPAT_GENERIC": "banded,full,traceback"}),
             ("sync_free+no_line_native", {"NVBIO_HIP_SYNC_FREE": "1", "NVBIO_HIP_COMPAT_LINE_NATIVE": "0"}),
             ("all_off", {"NVBIO_HIP_SYNC_FREE": "1", "NVBIO_HIP_COMPAT_LINE_NATIVE": "0", "NVBIO_HIP_COMPAT_GENERIC": "banded,full,traceback"})]
    if os.environ.get("MT_MATRIX_QUICK"):
        cases = quick
    for tag, env in cases:
        ref = st
        if "GENERIC" in "".join(env) or "NO_VIEWS" in "".join(env):
            ref = run("st_" + tag, False, env)            # (same results expected, but compare like with like)
            out[tag + "_single_equals_default_single"] = (ref == st) if not isinstance(ref, str) else ref
        mt = run("mt_" + tag, True, env)
        if isinstance(mt, str) or isinstance(ref, str):
            out[tag] = mt if isinstance(mt, str) else ref
        else:
            out[tag] = {"records_only_single": sum((ref - mt).values()), "records_only_two_threads": sum((mt - ref).values())}
        print(tag, out[tag], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
