#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection CSVs per (kernel, counter).
usage: python tools/pmc_summary.py <dir with p*/p*_counter_collection.csv>"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(sys.argv[1] + "/p*/p*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0][:64]
        if not name.startswith("dsa::") and "dsa" not in name:
            continue
        k = (name, row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"])
        acc[k][1] += 1
names = sorted({k[0] for k in acc})
for n in names:
    print(n)
    for (kn, c), (s, cnt) in sorted(acc.items()):
        if kn == n:
            print(f"    {c:28s} {s / cnt:18.1f}   (avg of {cnt} dispatches)")
