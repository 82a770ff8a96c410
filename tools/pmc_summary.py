#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per (kernel, grid) mean counters."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "dm::" not in name and "_ZN2dm" not in name:
            continue
        import re
        m = re.search(r"(\w+_kernel\w*(?:<[^>]*>)?)", name)          # keep template arguments: they tell the variants apart
        name = m.group(1) if m else name.split("(")[0][-60:]
        acc[(name, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
