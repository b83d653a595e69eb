#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean of every counter per (short) kernel name."""
import csv
import glob
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
only = sys.argv[2] if len(sys.argv) > 2 else None
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)(<[^>(]*>)?", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        if only and only not in name:
            continue
        key = only if only else name + " grid=" + r["Grid_Size"]       # every instantiation of the family in one row
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name[:100])
    print("    " + "  ".join("%s: n=%d mean=%.6g sum=%.6g" % (c.replace("SQ_", ""), len(v), sum(v) / len(v), sum(v))
                             for c, v in sorted(cs.items())))
