"""Per-kernel device time inside the tail of a rocprofv3 --kernel-trace CSV: the window that starts at the N-th from last
dispatch of a marker kernel (default map_exact_kernel) -- e.g. the C++ driver leg at the end of tools/cxx_leg_probe.py, whose
six batches of three seeding passes are the last 18 map_exact launches.  Prints the share of nvb:: kernels in that window.
usage: python tools/trace_window.py <kernel_trace.csv> [n_markers] [marker_substring]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    n_mark = int(sys.argv[2]) if len(sys.argv) > 2 else 18
    marker = sys.argv[3] if len(sys.argv) > 3 else "map_exact_kernel"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < n_mark:
        raise SystemExit("only %d dispatches of %s" % (len(marks), marker))
    win = rows[marks[-n_mark]:]
    tot = defaultdict(lambda: [0, 0])
    for s, e, name in win:
        short = name.split("(")[0]
        if len(short) > 90:
            short = short[:87] + "..."
        tot[short][0] += e - s; tot[short][1] += 1
    total = sum(v[0] for v in tot.values())
    ours = sum(v[0] for k, v in tot.items() if "nvb::" in k)
    print("window: %d dispatches, %.3f ms of kernel time over %.3f ms of wall time; nvb:: kernels %.1f %%" %
          (len(win), total / 1e6, (win[-1][1] - win[0][0]) / 1e6, 100.0 * ours / total))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:25]:
        print("%8.3f ms %5.1f %% %6d  %s" % (v[0] / 1e6, 100.0 * v[0] / total, v[1], k))


if __name__ == "__main__":
    main()
