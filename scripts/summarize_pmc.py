#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (sum and mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0) or 0)
    out = path.replace("counter_collection.csv", "pmc_summary.txt")
    with open(out, "w") as o:
        o.write("# %s\n# kernel, counter, dispatches, sum, mean_per_dispatch\n" % os.path.basename(path))
        for (kn, cn), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write("%s, %s, %d, %.6g, %.6g\n" % (kn, cn, n, s, s / max(n, 1)))
    print("==", out)
    print(open(out).read()[:3000])
