#!/usr/bin/env python3
"""Chronological runs of identical dispatches (same kernel, same grid) in a rocprofv3 counter_collection CSV with the average counter
value of each run: separates the shapes of a microbench pass that share one kernel symbol.  usage: pmc_runs.py <csv> [name filter]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1], newline="")), key=lambda r: int(r["Dispatch_Id"]))
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm_"
runs = []
for r in rows:
    if flt not in r["Kernel_Name"] and "splitk" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:60], r["Grid_Size"], r["Counter_Name"])
    if runs and runs[-1][0] == key:
        runs[-1][1].append(float(r["Counter_Value"]))
    else:
        runs.append((key, [float(r["Counter_Value"])]))
for (name, grid, ctr), v in runs:
    print(f"{name:60s} grid {grid:>8s} n={len(v):2d} {ctr} avg {sum(v) / len(v) * 1024 / 1e6:9.2f} MB (raw KB counter x 1024)")
