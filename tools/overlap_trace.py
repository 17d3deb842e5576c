#!/usr/bin/env python3
"""Concurrency inside a rocprofv3 --kernel-trace CSV: from the first dispatch of a marker kernel on, how long 0 / 1 / 2+ kernels were
resident at once, the summed kernel time against the wall time, and each kernel's mean duration -- to see whether two batches in
flight (tools/batch_probe.py) actually overlap the fabric-bound seeding kernels with the VALU-bound extension kernels.
usage: python tools/overlap_trace.py <kernel_trace.csv> [marker_substring]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "map_exact_kernel"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
    rows.sort()
    first = next(i for i, r in enumerate(rows) if marker in r[2])
    rows = rows[first:]
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    level, last, hist = 0, ev[0][0], defaultdict(int)
    for t, d in ev:
        hist[min(level, 3)] += t - last
        last = t; level += d
    wall = rows and (max(r[1] for r in rows) - rows[0][0]) or 0
    ksum = sum(e - s for s, e, _, _ in rows)
    print("dispatches %d, queues %d; wall %.2f ms, summed kernel time %.2f ms (%.2fx)" % (len(rows), len(set(r[3] for r in rows)), wall / 1e6, ksum / 1e6, ksum / max(wall, 1)))
    print("time with 0 / 1 / 2 / 3+ kernels resident: %s ms" % " / ".join("%.2f" % (hist[k] / 1e6) for k in range(4)))
    # which kernels overlap which: time kernel A spends while a kernel of another queue is resident
    tot = defaultdict(lambda: [0, 0, 0])
    active = []
    for s, e, name, q in rows:
        tot[name][0] += e - s; tot[name][1] += 1
    # overlap per kernel with any other kernel (sweep)
    import bisect
    starts = [r[0] for r in rows]
    for i, (s, e, name, q) in enumerate(rows):
        ov = 0
        j = i - 1
        while j >= 0 and i - j < 64:
            s2, e2 = rows[j][0], rows[j][1]
            if e2 > s:
                ov += min(e, e2) - s
            j -= 1
        j = i + 1
        while j < len(rows) and rows[j][0] < e:
            ov += min(e, rows[j][1]) - rows[j][0]
            j += 1
        tot[name][2] += min(ov, e - s)
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
        print("%9.3f ms total %6d calls  mean %8.3f ms  overlapped %5.1f %%  %s" % (v[0] / 1e6, v[1], v[0] / v[1] / 1e6, 100.0 * v[2] / max(v[0], 1), k[:80]))


if __name__ == "__main__":
    main()
