#!/usr/bin/env python3
"""rocprofv3 counter CSVs -> per-dispatch values in dispatch order, grouped in runs of `n` launches (dev A/B where the variants
share one kernel symbol and only the launch order tells them apart).  usage: pmc_by_order.py <dir> <kernel substring> <n per group>"""
import csv, glob, sys
from collections import defaultdict
d, sub, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"], float(r["Counter_Value"])))
rows.sort()
by = defaultdict(list)
for did, k, c, v in rows:
    by[c].append((did, k, v))
for c, lst in by.items():
    print(c)
    for g in range(0, len(lst), n):
        grp = lst[g:g + n]
        print(f"  launches {grp[0][0]}..{grp[-1][0]} {grp[0][1]}: " + " ".join(f"{v / 1e6:.3f}" for _, _, v in grp) + "  (1e6 KB-units)")
