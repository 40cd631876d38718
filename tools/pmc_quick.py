#!/usr/bin/env python3
"""Average of every counter per kernel symbol in a rocprofv3 counter_collection CSV (quick look; tools/pmc_summary.py is the full summary)."""
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for path in glob.glob(f, recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: (round(sum(v) / len(v) / 1e6, 2), len(v)) for c, v in d.items()})
