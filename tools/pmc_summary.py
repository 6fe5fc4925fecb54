#!/usr/bin/env python3
"""Average per-dispatch PMC values for kernels whose name contains a substring (rocprofv3 --pmc csv output)."""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(agg):
        print(f"{k:32s} {agg[k]/n[k]:16.1f}  (x{n[k]})")
