#!/usr/bin/env python3
"""Per-kernel PMC averages from rocprofv3 `--pmc ... --output-format csv` runs.
usage: tools/pmc_summary.py <dir with pmc_*/p_counter_collection.csv> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_packet"
res = {}
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "*counter_collection.csv"))):
    acc = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        if want not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for c, per in acc.items():
        vals = list(per.values())
        res[c] = (sum(vals) / len(vals), len(vals))
for c, (v, n) in res.items():
    print(f"{c:28s} {v:20.1f}  (avg over {n} dispatches)")
