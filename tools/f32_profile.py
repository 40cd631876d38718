#!/usr/bin/env python3
"""fp32 engine (SDE_F32) at SD v1.5 size: wall time of a UNet batch-2 forward (one CFG step of BASELINE configs[0]) and the engine
profiler's per-class table.   python tools/f32_profile.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from diffusers_amd import modelgen
from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline

MODEL_DIR = os.environ.get("SDE_BENCH_MODEL_DIR", "/tmp/sde_bench_sd15_f16")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    if not os.path.exists(os.path.join(MODEL_DIR, ".complete")):
        modelgen.write_model_dir(MODEL_DIR, modelgen.unet_config(), modelgen.vae_config(), seed=42, dtype=np.float16)
        open(os.path.join(MODEL_DIR, ".complete"), "w").write("ok")
    p = StableDiffusionPipeline(MODEL_DIR, StableDiffusionOptions(dtype="f32"), load_vae=False)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 4, 64, 64)).astype(np.float32)
    ehs = (rng.standard_normal((2, 77, 768)) * 0.5).astype(np.float32)
    p.unet(x, 500.0, ehs)
    t0 = time.perf_counter()
    for _ in range(reps):
        p.unet(x, 500.0, ehs)
    wall = (time.perf_counter() - t0) / reps * 1e3
    p.profile_reset(); p.profile(True)
    p.unet(x, 500.0, ehs)
    st = sorted(p.profile_get(), key=lambda s: -s["total_ms"])
    p.profile(False)
    tot = sum(s["total_ms"] for s in st)
    print(f"fp32 engine, SD1.5 UNet batch 2, 64x64 latents: {wall:.1f} ms per forward (host wall, H2D/D2H included); profiled kernels {tot:.1f} ms")
    for s in st[:14]:
        tf = s["flops"] / s["total_ms"] / 1e9 if s["flops"] else 0.0
        print(f"  {s['name']:28s} {s['launches']:5d} launches {s['total_ms']:9.2f} ms  {tf:7.1f} TFLOP/s", flush=True)
    p.close()


if __name__ == "__main__":
    main()
