#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: one line per (kernel, counter) with mean values."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("(")[0][-60:]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for name, cs in acc.items():
    d = list(dur[name].values())
    print("%s  launches=%d  mean_ms=%.4f" % (name, len(d), sum(d) / len(d)))
    for c, v in sorted(cs.items()):
        print("    %-28s %.6g" % (c, sum(v) / len(v)))
