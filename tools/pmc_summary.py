#!/usr/bin/env python3
"""Per-kernel-class summary of a `rocprofv3 --pmc ... --kernel-trace --output-format csv` counter_collection CSV.

usage: python tools/pmc_summary.py <counter_collection.csv> [out.json]
       python tools/pmc_summary.py --traffic <FETCH_SIZE pass csv> <WRITE_SIZE pass csv> <out.json> [source note] [unet_weight]

--traffic with unet_weight W (e.g. 15): the passes ran FEWER denoise steps than the bench (2 instead of 30: a 30-step run under --pmc
crashed inside the profiler, gpurun_out/pmc_f.log), so every dispatch up to the last cfg_scheduler_step launch (the denoise loop) is
counted W times, bytes and launches alike, and the VAE decode after it once: per-launch averages of a kernel class are then those of
the bench's own launch mix (UNet : VAE launches = 30 : 1).  Shapes are deterministic, so this is exact, not an estimate.  "phases" in
the output keeps the un-mixed per-phase figures.

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
    if base == "gemm_midl_kernel":      # loader-wave variant: the 3x3 convolutions of the 256x160 class; <.., 128>: the 128x160 linears, class gemm_mfma_ldw128
        return ("conv3x3_mfma_" if t and t[0] == "true" else "gemm_mfma_") + ("ldw128" if len(t) > 3 and t[3] == "128" else "256x160")
    if base == "gemm_mid_kernel":
        return ("conv3x3_mfma_" if t and t[0] == "true" else "gemm_mfma_") + "256x160"
    if base == "gemm_midh_kernel":
        return "conv3x3_mfma_halo"
    if base == "gemm_ws_kernel":          # <KT, NT, MT, EPI, HAS_R, WAVES>: EPI 2 = the GEGLU projection
        return "gemm_geglu_mfma_ws" if len(t) > 3 and t[3] == "2" else "gemm_mfma_ws"
    if base == "gemm_mfma_kernel":
        conv = len(t) > 2 and t[2] == "true"
        return ("conv3x3_mfma_" if conv else "gemm_mfma_") + f"{t[0]}x{t[1]}" if len(t) > 1 else "gemm_mfma"
    if base == "attn_mfma_kernel":
        return f"attn_mfma<{t[0]},{t[1]}>" if len(t) > 1 else "attn_mfma"
    return base


def symbol_key(name: str) -> str:
    """The kernel symbol as bench.py's ROCPROF_SYMBOL table spells it (template arguments that only select a loader variant folded)."""
    m = re.match(r"(?:void )?(\w+)(?:<(.*)>)?", name)
    base, targs = (m.group(1), m.group(2) or "") if m else (name, "")
    t = [a.strip() for a in targs.split(",")] if targs else []
    if base == "gemm_big_kernel" and len(t) >= 4:
        return f"gemm_big_kernel<{t[0]}, {'UPS' if t[0] == 'true' else t[1]}, {t[2]}, {t[3]}, LO>"
    if base == "gemm_midl_kernel" and len(t) >= 2:
        return f"gemm_midl_kernel<{t[0]}, {'UPS' if t[0] == 'true' else t[1]}, LO, {t[3] if len(t) > 3 else '256'}>"
    if base == "gemm_mid_kernel" and len(t) >= 3:
        return f"gemm_mid_kernel<{t[0]}, {'UPS' if t[0] == 'true' else t[1]}, {t[2]}, LO>"
    if base == "gemm_midh_kernel":
        return "gemm_midh_kernel<LO, UPS, NT>"
    if base == "gemm_ws_kernel":
        return "gemm_ws_kernel<5, NT, MT, EPI, R, WAVES>"
    if base == "gemm_mfma_kernel" and len(t) >= 3:
        return f"gemm_mfma_kernel<BM, BN, {t[2]}>"
    if base == "attn_mfma_kernel" and len(t) >= 2:
        return f"attn_mfma_kernel<{t[0]}, {t[1]}, ONES, FOLD>"
    return base


def engine_class(name: str, dur_ns: float, split: dict) -> str:
    """Engine-profiler class name (what bench.py's kernel table uses)."""
    k = kernel_class(name)
    if k.startswith("conv3x3_mfma_") or k.startswith("gemm_mfma_"):
        base, tile = k.rsplit("_", 1)
        return base if tile in ("128x160", "64x160", "128x128", "64x128") else k
    m = re.search(r"(gn_stats|gn_apply|gn_small|layernorm|cfg_step|splitk_reduce|gemv)", name)
    return {"gn_stats": "groupnorm_stats", "gn_apply": "groupnorm_apply", "gn_small": "groupnorm_fused_small", "layernorm": "layernorm",
            "cfg_step": "cfg_scheduler_step", "splitk_reduce": "splitk_reduce", "gemv": "gemv_time_mlp"}.get(m.group(1), k) if m else k


def traffic(fetch_csv: str, write_csv: str, out_path: str, source: str, unet_weight: float = 1.0):
    """profiles/r*_pmc_traffic.json: HBM-side bytes per launch per engine kernel class from two separate --pmc passes."""
    res = defaultdict(lambda: defaultdict(float))
    sym = defaultdict(lambda: defaultdict(float))
    phase = defaultdict(lambda: defaultdict(float))
    for path, counter, key in ((fetch_csv, "FETCH_SIZE", "fetch_kb"), (write_csv, "WRITE_SIZE", "write_kb")):
        rows = [r for r in csv.DictReader(open(path, newline="")) if r["Counter_Name"] == counter]
        loop_end = max((int(r["Dispatch_Id"]) for r in rows if "cfg_step" in r["Kernel_Name"]), default=-1)
        split = {}
        for r in rows:
            k, d = kernel_class(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            lo, hi = split.get(k, (d, d))
            split[k] = (min(lo, d), max(hi, d))
        for r in rows:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            in_loop = int(r["Dispatch_Id"]) <= loop_end
            wgt = unet_weight if in_loop else 1.0
            e = res[engine_class(r["Kernel_Name"], d, split)]
            e[key] += wgt * float(r["Counter_Value"])
            e[key + "_n"] += wgt
            g = sym[symbol_key(r["Kernel_Name"])]
            g[key] += wgt * float(r["Counter_Value"])
            g[key + "_n"] += wgt
            h = phase[symbol_key(r["Kernel_Name"]) + (" [denoise loop]" if in_loop else " [vae decode]")]
            h[key] += float(r["Counter_Value"])
            h[key + "_n"] += 1
    out = {"source": source, "kernels": {}, "symbols": {}, "phases": {}}
    for k, e in sorted(phase.items()):
        if e.get("fetch_kb_n") and e.get("write_kb_n"):
            out["phases"][k] = {"launches_in_pass": int(e["fetch_kb_n"]),
                                "fetch_MB_per_launch_corrected_x2": round(2 * e["fetch_kb"] * 1024 / e["fetch_kb_n"] / 1e6, 2),
                                "write_MB_per_launch": round(e["write_kb"] * 1024 / e["write_kb_n"] / 1e6, 2)}
    for k, e in sorted(sym.items(), key=lambda kv: -(kv[1]["fetch_kb"] + kv[1]["write_kb"])):
        if e.get("fetch_kb_n") and e.get("write_kb_n"):
            f = e["fetch_kb"] * 1024 / e["fetch_kb_n"] / 1e6
            out["symbols"][k] = {"launches": int(e["fetch_kb_n"]), "fetch_MB_per_launch_raw": round(f, 2),
                                 "fetch_MB_per_launch_corrected_x2": round(2 * f, 2),
                                 "write_MB_per_launch": round(e["write_kb"] * 1024 / e["write_kb_n"] / 1e6, 2)}
    for k, e in sorted(res.items(), key=lambda kv: -(kv[1]["fetch_kb"] + kv[1]["write_kb"])):
        if not e.get("fetch_kb_n") or not e.get("write_kb_n"):
            continue
        f = e["fetch_kb"] * 1024 / e["fetch_kb_n"] / 1e6
        out["kernels"][k] = {"launches": int(e["fetch_kb_n"]), "fetch_MB_per_launch_raw": round(f, 2),
                             "fetch_MB_per_launch_corrected_x2": round(2 * f, 2),
                             "write_MB_per_launch": round(e["write_kb"] * 1024 / e["write_kb_n"] / 1e6, 2)}
    open(out_path, "w").write(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


def main():
    if sys.argv[1] == "--traffic":
        return traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "", float(sys.argv[6]) if len(sys.argv) > 6 else 1.0)
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
