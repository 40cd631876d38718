#!/usr/bin/env python3
"""Per-kernel-class summary of a `rocprofv3 --pmc ... --kernel-trace --output-format csv` counter_collection CSV.

usage: python tools/pmc_summary.py <counter_collection.csv> [out.json]

Counters are summed per kernel class over all dispatches; durations are End-Start of the same dispatches (under PMC
collection, i.e. slower than un-profiled).  Derived figures:

  mfma_util_pct      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/XCC * SIMD_NUM) * 100   (rocprofv3's MfmaUtil formula with the
                       per-XCD reduce(max) replaced by sum/8: the CSV holds the sum over the 8 XCDs; SIMD_NUM = 1024)
  mfma_tflops        = SQ_INSTS_VALU_MFMA_MOPS_F16 * 512 / time          (flops actually issued to the matrix pipe, padding included)
  clock_ghz          = GRBM_GUI_ACTIVE / 8 / time
  lds_conflict_pct   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE * 100
"""
import csv
import json
import re
import sys
from collections import defaultdict

XCC, SIMD_NUM = 8, 1024


def kernel_class(name: str) -> str:
    """Same class names as the engine's own profiler (kernels.h / bench.py)."""
    m = re.match(r"(?:void )?(\w+)(?:<(.*)>)?", name)
    base, targs = (m.group(1), m.group(2) or "") if m else (name, "")
    t = [a.strip() for a in targs.split(",")] if targs else []
    if base == "gemm_big_kernel":
        conv = t[0] == "true"
        nt = t[2] if len(t) > 2 else "5"
        tile = "256x320" if nt == "5" else "256x256"
        return ("conv3x3_mfma_" if conv else "gemm_mfma_") + tile
    if base == "gemm_mid_kernel":
        return ("conv3x3_mfma_" if t and t[0] == "true" else "gemm_mfma_") + "256x160"
    if base == "gemm_mfma_kernel":
        conv = len(t) > 2 and t[2] == "true"
        return ("conv3x3_mfma_" if conv else "gemm_mfma_") + f"{t[0]}x{t[1]}" if len(t) > 1 else "gemm_mfma"
    if base == "attn_mfma_kernel":
        return f"attn_mfma<dqk={t[0]},dvt={t[1]}>" if len(t) > 1 else "attn_mfma"
    return base


def main():
    path = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(float))
    seen = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = kernel_class(row["Kernel_Name"])
            a = agg[k]
            a[row["Counter_Name"]] += float(row["Counter_Value"])
            d = row["Dispatch_Id"]
            if d not in seen:
                seen.add(d)
                a["_ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                a["_n"] += 1
            if row["Counter_Name"] == "MfmaUtil":
                a["_mfmautil_x_ns"] += float(row["Counter_Value"]) * (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"]):
        ns = a["_ns"]
        if ns <= 0:
            continue
        e = {"launches": int(a["_n"]), "total_ms_under_pmc": round(ns / 1e6, 3)}
        gui = a.get("GRBM_GUI_ACTIVE", 0.0)
        if gui:
            e["clock_ghz"] = round(gui / XCC / ns, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a and gui:
            e["mfma_util_pct"] = round(100.0 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / XCC * SIMD_NUM), 2)
        if "_mfmautil_x_ns" in a:
            e["mfma_util_pct_rocprofv3_derived"] = round(a["_mfmautil_x_ns"] / ns, 2)
        if "SQ_INSTS_VALU_MFMA_MOPS_F16" in a:
            e["mfma_tflops_issued"] = round(a["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512 / ns / 1e3, 1)
        if a.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_pct"] = round(100.0 * a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"], 2)
        for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_INST_ANY"):
            if c in a:
                e[c] = a[c]
        out[k] = e
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
