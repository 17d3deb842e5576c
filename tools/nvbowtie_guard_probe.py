#!/usr/bin/env python3
"""Does anything under the unchanged nvBowtie write outside its device buffers?  Runs the application (single-thread and two-thread mode, tuned and generic
routes of the drop-in layer) with NVBIO_HIP_POOL_GUARD: every pool block gets guard zones on both sides, checked at its free, periodically and at exit.
Files: a `tools/nvbowtie_3gbp.py --keep DIR` run.  GPU box only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    W = sys.argv[1]
    extra = sys.argv[2].split() if len(sys.argv) > 2 else ["--batch-size", "64"]
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    gen = {"NVBIO_HIP_COMPAT_GENERIC": "banded,full,traceback"}
    out = {}
    for tag, mt, env in [("mt_generic", True, gen), ("mt_generic_2", True, gen), ("mt_generic_3", True, gen), ("mt_generic_4", True, gen), ("mt", True, {}), ("mt_2", True, {}), ("mt_3", True, {})]:
        sam = os.path.join(W, "guard_" + tag + ".sam")
        try:
            r = subprocess.run([exe] + (["--device", "0", "--device", "0"] if mt else []) + extra + ["--file-ref", "-x", os.path.join(W, "genome"), "-U", os.path.join(W, "reads.fastq"), "-S", sam],
                               capture_output=True, text=True, timeout=200, env=dict(os.environ, NVBIO_HIP_POOL_GUARD="1024", **env))
        except subprocess.TimeoutExpired:
            out[tag] = "HUNG"; continue
        lines = [l for l in (r.stdout + r.stderr).replace("\r", "\n").splitlines() if "pool guard" in l or "live block" in l]
        out[tag] = dict(exit=r.returncode, guard_reports=len(lines), first=lines[:40])
        print(tag, r.returncode, len(lines), file=sys.stderr, flush=True)
        for l in lines[:40]:
            print("   ", l, file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
