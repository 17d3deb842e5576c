#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: one line per (kernel, counter) with mean values."""
import csv
import sys
from collections import defaultdict

each = "--each" in sys.argv          # one block per dispatch (in dispatch order) instead of the mean over a kernel's launches
args = [a for a in sys.argv[1:] if a != "--each"]
path, pat = args[0], (args[1] if len(args) > 1 else "")
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("(")[0][-60:]
        if each:
            name = "%s #%06d" % (name, int(r["Dispatch_Id"]))
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for name, cs in sorted(acc.items()):
    d = list(dur[name].values())
    print("%s  launches=%d  mean_ms=%.4f" % (name, len(d), sum(d) / len(d)))
    for c, v in sorted(cs.items()):
        print("    %-28s %.6g" % (c, sum(v) / len(v)))
