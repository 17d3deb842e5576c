#!/usr/bin/env python3
"""One read's hit selection, pick by pick, in the unchanged nvBowtie (a build with the reference's own device-side debug prints switched on:
NVBOWTIE_DEBUG_BUILD=1 tools/nvbowtie_tu_check.py --link -> oracle/_ref/ref_nvBowtie_dbg, --debug-read K --debug-select 1) and in this
repository's driver (nvbio_amd.aligner.TRACE), on the 300-read subset of tools/nvbowtie_3gbp_probe_ids.txt.  GPU box only."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
W = sys.argv[1] if len(sys.argv) > 1 else "/tmp/w3g"
KS = [int(x) for x in sys.argv[2:]] or [0, 1]


def main():
    from nvbio_amd import io as nio, aligner as A
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie_dbg")
    raw = np.fromfile(W + "/reads.fastq", dtype=np.uint8).reshape(-1, 215)
    ids = [int(x) for x in open(os.path.join(ROOT, "tools", "nvbowtie_3gbp_probe_ids.txt")).read().split()] if os.environ.get("PROBE_ALL_READS") != "1" else list(range(min(300, raw.shape[0])))
    raw[ids].tofile(W + "/sub.fastq")
    lut = np.full(256, 4, np.uint8)
    for c, v in zip(b"ACGT", range(4)):
        lut[c] = v
    sym = torch.from_numpy(lut[raw[ids][:, 11:111]]).cuda(); qual = torch.from_numpy(raw[ids][:, 114:214] - 33).cuda()
    data = nio.FMIndexDataDevice(W + "/genome", flags=nio.FORWARD | nio.SA, device="cuda")
    n_genome, g_words = nio.load_genome(W + "/genome")
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).cuda()
    names = ["r%08d" % i for i in ids]
    m = len(ids)
    index = torch.arange(0, (m + 1) * 100, 100, dtype=torch.int64, device="cuda")
    out = {}
    for K in KS:
        try:
            r = subprocess.run([exe, "--no-multi-hits", "1", "--debug-read", str(K), "--debug-select", "1", "--file-ref", "-x", W + "/genome", "-U", W + "/sub.fastq", "-S", W + "/dbg.sam"],
                               capture_output=True, text=True, timeout=180)
            log = (r.stdout + r.stderr).replace("\r", "\n")
            picks = re.findall(r"selected hit\[(\d+)\], SA\[(\d+):(\d+):(\d+)\]", log)
            raw_lines = [l for l in log.splitlines() if "selected" in l][:400]
            other = [l for l in log.splitlines() if "select" in l and "selected hit" not in l][:10]
        except subprocess.TimeoutExpired:
            picks, other, raw_lines = "hung", [], []
        A.TRACE = {"read": K, "events": []}
        batch = A.ReadBatch.from_ragged(sym.reshape(-1), index, qual.reshape(-1))
        rr = A.best_approx(data.index(), data.rindex(), batch, genome_words, n_genome, A.Params(hits_stride=32, no_multi_hits=True), names=names, cigar_stride=64, finish=True)
        torch.cuda.synchronize()
        ev = A.TRACE["events"]; A.TRACE = None
        # the read's hits as this repository's mapper leaves them after the first seeding pass
        from nvbio_amd import mapping as MP
        hits, counts, _ = MP.map_seeds(data.index(), None, batch.reversed, A.Params(hits_stride=32).mapping_params(), 100, hits_stride=32)
        hk = MP.unpack_seed_hits(hits[K, :int(counts[K].item())])
        own_hits = [dict(begin=int(b), delta=int(d), pos=int(p_), rc=int(r_)) for b, d, p_, r_ in zip(hk["range_begin"].tolist(), hk["range_delta"].tolist(), hk["pos_in_read"].tolist(), hk["rc"].tolist())]
        own = [(e["sa_rows"][0], (e["seeds"][0]) & 0x3FF, (e["seeds"][0] >> 12) & 1, (e["seeds"][0] >> 13) & 1, e["positions"][0], e["scores"][0]) for e in ev if e["sa_rows"]]
        out[str(K)] = dict(read=names[K], own_hits_first_pass=own_hits, nvbowtie_raw_lines=raw_lines, nvbowtie_picks=[(int(h), int(sa), int(d), int(p)) for h, sa, d, p in picks] if picks != "hung" else "hung", nvbowtie_other=other,
                           own_picks_sa_pos_indexdir_rc_position_score=own,
                           first_divergence=next((i for i, (a, b) in enumerate(zip([int(p[1]) for p in picks], [o[0] for o in own])) if a != b), None) if picks != "hung" else None)
    print(json.dumps(out, indent=1, default=str))


if __name__ == "__main__":
    main()
