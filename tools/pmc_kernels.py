#!/usr/bin/env python3
"""Per-kernel averages of every counter of one or more rocprofv3 --pmc passes (directories given on the command line) for the kernels
whose name matches a pattern: what a kernel that sits at neither roofline is waiting for.  Usage:
pmc_kernels.py <name regex> <pass dir> [<pass dir> ...]   (each pass: --pmc ... --kernel-trace --output-format csv)"""
import collections
import csv
import glob
import os
import re
import sys

pat = re.compile(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for cc in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(cc)):
            n = r["Kernel_Name"]
            if pat.search(n):
                agg[re.sub(r"\(.*", "", n)[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for tr in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(tr)):
            n = r["Kernel_Name"]
            if pat.search(n):
                agg[re.sub(r"\(.*", "", n)[:60]]["duration_us (pass %s)" % os.path.basename(d.rstrip("/"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:34s} {sum(v) / len(v):16.1f}   (n={len(v)})")
