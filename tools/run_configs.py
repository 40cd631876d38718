#!/usr/bin/env python3
"""Runs BASELINE.json configs[1], [3] and [4] (and the single-GPU share of [2]) at FULL size on one MI355X and prints one JSON
line per config: wall time, images/s, UNet it/s.  Synthetic weights and embeddings (no checkpoints in the container).

    python tools/run_configs.py [--only 2,4,5] [--out profiles/r1_configs.json]

configs (1-based, as in BASELINE.json):
  1  SD1.5 512x512, Euler 20 steps, CFG 7.5, batch 1, fp32 ENGINE (configs[0], the reference's CPU-runnable case; --only 1,...)
  2  SD1.5 512x512, DPM++ 30 steps, CFG 7.5, batch 8                      (the bench.py workload)
  3  SD1.5 512x512, Euler-a 30 steps, micro-batch 8 of the 256 prompts    (one rank's share; sharding is bench.py --gpus N)
  4  SD2.1 768x768 v-prediction, DDIM "50 steps" (49 UNet calls, reference quirk), batch 4, heads of d=64, linear projections
  5  SD1.5 512x512, LMS 25 steps, batch 16, text length 227 (weighted long prompt), + 16 VAE decodes
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def model_dir(kind):
    from diffusers_amd import modelgen
    path = f"/tmp/sde_cfg_{kind}_f16"
    if not os.path.exists(os.path.join(path, ".complete")):
        t0 = time.time()
        ucfg = modelgen.unet_config() if kind == "sd15" else modelgen.sd21_unet_config()
        modelgen.write_model_dir(path, ucfg, modelgen.vae_config(), seed=42, dtype=np.float16, unet_gain=1.0)
        open(os.path.join(path, ".complete"), "w").write("ok")
        print(f"[configs] wrote {path} in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
    return path


CONFIGS = {
    1: dict(model="sd15", H=512, W=512, B=1, S=77, D=768, sched="euler", steps=20, guidance=7.5, pred=None, dtype="f32"),   # configs[0]: the CPU-runnable case, here on the fp32 ENGINE
    2: dict(model="sd15", H=512, W=512, B=8, S=77, D=768, sched="dpmpp", steps=30, guidance=7.5, pred=None),
    3: dict(model="sd15", H=512, W=512, B=8, S=77, D=768, sched="euler_a", steps=30, guidance=7.5, pred=None),
    4: dict(model="sd21", H=768, W=768, B=4, S=77, D=1024, sched="ddim", steps=50, guidance=7.5, pred="v"),
    5: dict(model="sd15", H=512, W=512, B=16, S=227, D=768, sched="lms", steps=25, guidance=7.5, pred=None),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="2,3,4,5")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline, StableDiffusionTxt2ImgOptions
    dev = torch.device("cuda", 0)
    results = []
    pipes = {}
    for idx in [int(x) for x in args.only.split(",")]:
        c = CONFIGS[idx]
        key = (c["model"], c.get("dtype", "f16"))
        if key not in pipes:
            for p in pipes.values():
                p.close()
            pipes.clear()
            pipes[key] = StableDiffusionPipeline(model_dir(c["model"]), StableDiffusionOptions(dtype=key[1]))
        pipe = pipes[key]
        B, S, D, H, W = c["B"], c["S"], c["D"], c["H"], c["W"]
        g = torch.Generator(device="cpu").manual_seed(idx)
        ehs = (torch.randn(2 * B, S, D, generator=g) * 0.5).to(dev)
        init = torch.randn(B, 4, H // 8, W // 8, generator=g).to(dev)
        images = torch.empty(B, H, W, 3, device=dev, dtype=torch.float32)
        ptrs = dict(ehs=ehs.data_ptr(), S=S, D=D, init_latents=init.data_ptr(), images=images.data_ptr())
        opts = StableDiffusionTxt2ImgOptions(height=H, width=W, guidance_scale=c["guidance"], steps=c["steps"], batch_size=B, seed=idx)

        def sched():
            cls = N.BY_NAME[c["sched"]]
            if c["pred"] == "v":
                return cls.stable_diffusion_v1_optimized_default(prediction_type="v_prediction")
            return cls.stable_diffusion_v1_optimized_default()

        best = None
        for it in range(args.repeat + 1):          # first pass = warm-up (arena sizing, K/V cache)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = opts.run(pipe, sched(), device_ptrs=ptrs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"[configs] config {idx} pass {it}: {dt * 1e3:.1f} ms ({r.unet_calls} UNet calls)", file=sys.stderr, flush=True)
            if it and (best is None or dt < best[0]):
                best = (dt, r)
        assert torch.isfinite(images).all(), f"config {idx}: non-finite pixels"
        dt, r = best
        line = {"config": idx, "workload": c, "images": B, "seconds": round(dt, 4), "images_per_s": round(B / dt, 3),
                "unet_calls": r.unet_calls, "unet_ms_per_call": round(r.unet_ms / r.unet_calls, 3),
                "unet_it_per_s": round(r.unet_calls / (r.unet_ms * 1e-3), 2), "vae_ms": round(r.vae_ms, 2),
                "image_mean": round(float(images.mean()), 4), "image_std": round(float(images.std()), 4)}
        print(json.dumps(line), flush=True)
        results.append(line)
    if args.out:
        json.dump({"device": torch.cuda.get_device_name(0), "note": "one MI355X, fp16 engine (config 1: fp32 engine), synthetic weights; best of "
                   f"{args.repeat} after one warm-up pass; inputs resident in HBM", "results": results}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
