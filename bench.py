#!/usr/bin/env python3
"""bench.py — images/sec of the txt2img hot path on MI355X (BASELINE.json metric).

A "step" is one full `StableDiffusionTxt2ImgOptions::run` over one batch: B=8 prompts, SD v1.5 geometry, 512x512,
DPM-Solver++ 30 steps, CFG 7.5 (UNet batch 16), fp16, followed by the batched VAE decode (BASELINE.json configs[1]).
Weights are random-init (no checkpoints exist offline), written once as a pyke model directory and loaded through the
product's own ONNX loader.  Text embeddings, initial latents and the image output buffer are resident in HBM before
the timed region.  With --gpus N every rank runs the same workload on its own GPU (independent prompts: no data-path
collective), value = all images / max-over-ranks time.
"""
import argparse
import ctypes as C
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

UNET_GFLOP_PER_SAMPLE = 803.3   # SD1.5 UNet forward, 64x64 latent, S=77
VAE_GFLOP_PER_IMAGE = 2514.5
PEAK_TFLOPS_F16 = 2500.0        # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def ensure_model_dir(path, rank, world, barrier):
    from diffusers_amd import modelgen
    marker = os.path.join(path, ".complete")
    if rank == 0 and not os.path.exists(marker):
        t0 = time.time()
        modelgen.write_model_dir(path, modelgen.unet_config(), modelgen.vae_config(), seed=42, dtype=np.float16, unet_gain=1.0)
        open(marker, "w").write("ok")
        print(f"[bench] wrote synthetic SD1.5 model dir {path} in {time.time() - t0:.1f}s", file=sys.stderr)
    barrier()
    return path


def cpu_baseline(model_dir, seconds_budget=30.0):
    """Oracle (torch CPU fp32 restatement of the reference's ORT graphs) on config[0]: B=1, CFG (UNet batch 2),
    Euler 20 steps, 512x512, fp32.  Bounded sample: UNet forwards until ~budget, + one VAE decode; extrapolated."""
    from oracle import model_dir as md, unet_ref
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    uw, ucfg = md.load_unet(model_dir)
    uw = {k: torch.from_numpy(v) for k, v in uw.items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 64, 64, generator=g) * 14.6
    ehs = torch.randn(2, 77, 768, generator=g) * 0.5
    with torch.no_grad():
        t0 = time.time()
        unet_ref.unet_forward(uw, ucfg, x, 999.0, ehs)   # warm-up
        warm = time.time() - t0
        n, t_un = 0, 0.0
        while n < 2 or (t_un + warm < seconds_budget * 0.6 and n < 4):
            t0 = time.time()
            unet_ref.unet_forward(uw, ucfg, x, 946.0, ehs)
            t_un += time.time() - t0
            n += 1
        t_unet = t_un / n
        del uw
        vw, vcfg = md.load_vae_decoder(model_dir)
        vw = {k: torch.from_numpy(v) for k, v in vw.items()}
        t0 = time.time()
        unet_ref.decode_latents(vw, vcfg, torch.randn(1, 4, 64, 64, generator=g) * 0.18215)
        t_vae = time.time() - t0
    s_per_image = 20 * t_unet + t_vae
    return {"value": 1.0 / s_per_image, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} timed UNet forwards (batch 2, fp32, 64x64 latent, S=77) + 1 VAE decode of one 512x512 image, "
                      f"extrapolated to Euler 20 steps (configs[0]); torch CPU oracle, ORT unavailable",
            "unet_s_per_call": t_unet, "vae_s_per_image": t_vae, "unet_it_per_s": 1.0 / t_unet}


# engine profiler class -> HIP kernel symbol(s) in a rocprofv3 kernel trace
ROCPROF_SYMBOL = {
    "conv3x3_mfma_256x320": "gemm_big_kernel<true, UPS, 5, 0>", "conv3x3_mfma_256x256": "gemm_big_kernel<true, UPS, 4, 0>",
    "gemm_mfma_256x320": "gemm_big_kernel<false, false, 5, 0> (shared with gemm_geglu_mfma_256x320)",
    "gemm_geglu_mfma_256x320": "gemm_big_kernel<false, false, 5, 0> (shared with gemm_mfma_256x320)",
    "gemm_mfma_256x256": "gemm_big_kernel<false, false, 4, 0>",
    "conv3x3_mfma_256x160": "gemm_midl_kernel<true, UPS>", "gemm_mfma_256x160": "gemm_mid_kernel<false, false, 0>",
    "conv3x3_mfma": "gemm_mfma_kernel<BM, BN, true>", "gemm_mfma": "gemm_mfma_kernel<BM, BN, false>",
    "splitk_reduce": "splitk_reduce_kernel",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--denoise-steps", type=int, default=30)
    ap.add_argument("--scheduler", default="dpmpp")
    ap.add_argument("--model-dir", default=os.environ.get("SDE_BENCH_MODEL_DIR", "/tmp/sde_bench_sd15_f16"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=25.0)
    ap.add_argument("--cfg-pair-split", action="store_true",
                    help="BASELINE config 3 variant: ranks (2p, 2p+1) share 2*batch prompts; each runs one CFG half of every UNet "
                         "call (UNet batch stays 2*batch per GPU) and they all-gather the halves over RCCL once per step")
    args = ap.parse_args()

    if os.environ.get("SDE_BENCH_DEBUG"):      # dump every thread's Python stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SDE_BENCH_DEBUG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        # SDE_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks then share devices)
        backend = os.environ.get("SDE_BENCH_BACKEND", "nccl")
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    model_dir = ensure_model_dir(args.model_dir, rank, world, barrier)

    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline, StableDiffusionTxt2ImgOptions

    t0 = time.time()
    pipe = StableDiffusionPipeline(model_dir, StableDiffusionOptions(device_id=local_rank))
    load_s = time.time() - t0
    B, S, D, H, W = args.batch, 77, 768, 512, 512
    pair = None
    if args.cfg_pair_split:
        from diffusers_amd.sharding import CfgPair, pair_layout
        assert dist is not None, "--cfg-pair-split needs an even number of ranks >= 2"
        pidx, half, n_pairs = pair_layout(world, rank)
        groups = [dist.new_group([2 * p, 2 * p + 1]) for p in range(n_pairs)]
        pair = CfgPair(dist, groups[pidx], half, dev)
        B = 2 * B                                      # prompts per PAIR; each GPU still runs a UNet batch of 2*args.batch
    g = torch.Generator(device="cpu").manual_seed(1 + (rank // 2 if pair else rank))   # identical inputs inside a pair
    ehs = (torch.randn(2 * B, S, D, generator=g) * 0.5).to(dev)
    init = torch.randn(B, 4, H // 8, W // 8, generator=g).to(dev)
    images = torch.empty(B, H, W, 3, device=dev, dtype=torch.float32)
    ptrs = dict(ehs=ehs.data_ptr(), S=S, D=D, init_latents=init.data_ptr(), images=images.data_ptr())
    opts = StableDiffusionTxt2ImgOptions(height=H, width=W, guidance_scale=7.5, steps=args.denoise_steps, batch_size=B, seed=0)

    def one_step():
        sch = N.BY_NAME[args.scheduler].stable_diffusion_v1_optimized_default()
        return opts.run(pipe, sch, device_ptrs=ptrs, cfg_pair=pair)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    unet_ms = vae_ms = enq_ms = 0.0
    calls = 0
    for _ in range(args.steps):
        r = one_step()
        unet_ms += r.unet_ms; vae_ms += r.vae_ms; calls += r.unet_calls; enq_ms += r.host_enqueue_ms
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mine = images if pair is None else (images[:(B + 1) // 2] if pair.half == 0 else images[(B + 1) // 2:])
    assert torch.isfinite(mine).all(), "non-finite pixels in the decoded images"

    # per-kernel timing: same workload once more with HIP events around every launch (on the engine's stream)
    roof = None
    kernel_table = []
    if pair is not None and rank != 0:
        one_step()                      # the pair's peer (and every other pair, to keep ranks aligned) takes part in the profiled pass
    if rank == 0:
        pipe.profile_reset(); pipe.profile(True)
        rp = one_step()
        stats = pipe.profile_get()
        pipe.profile(False)
        tot = sum(s["total_ms"] for s in stats) or 1.0
        for s in sorted(stats, key=lambda s: -s["total_ms"]):
            avg = s["total_ms"] / max(1, s["launches"])
            kernel_table.append(dict(name=s["name"], launches=s["launches"], total_ms=round(s["total_ms"], 3), avg_us=round(avg * 1e3, 2),
                                     share=round(s["total_ms"] / tot, 4),
                                     tflops=round(s["flops"] / (s["total_ms"] * 1e-3) / 1e12, 1) if s["flops"] and s["total_ms"] else None,
                                     gbs=round(s["bytes"] / (s["total_ms"] * 1e-3) / 1e9, 1) if s["bytes"] and s["total_ms"] else None))
        dom = max(stats, key=lambda s: s["total_ms"])
        # HBM-side bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the
        # figure comes from the committed rocprofv3 --pmc passes over this same workload (profiles/r1_pmc_traffic.json)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))["kernels"]
            for key, v in pmc.items():
                if dom["name"] in key.replace("(", "+").split("+"):
                    traffic = round((v["fetch_MB_per_launch_corrected_x2"] + v["write_MB_per_launch"]) * 1e6)
        except Exception:
            traffic = None
        if dom["flops"] > 0:
            ach = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
            # the HIP kernel symbol(s) behind the class, as `rocprofv3 --kernel-trace --stats` prints them (profiles/*kernel_stats*.csv)
            sym = ROCPROF_SYMBOL.get(dom["name"]) or re.sub(r"^attn_mfma<(\d+),(\d+)>$", r"attn_mfma_kernel<\1, \2, ONES, FOLD>", dom["name"])
            roof = {"kernel": dom["name"], "rocprof_symbol": sym, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS_F16, 4), "traffic": traffic, "launches": dom["launches"],
                    "avg_launch_us": round(dom["total_ms"] / dom["launches"] * 1e3, 2),
                    "flop_per_launch": dom["flops"] / dom["launches"], "profiled_step_ms": round(rp.total_ms, 2)}
        else:
            ach = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9
            roof = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic, "launches": dom["launches"],
                    "avg_launch_us": round(dom["total_ms"] / dom["launches"] * 1e3, 2)}

    total_images = (world // 2 if pair else world) * B * args.steps
    value = total_images / elapsed
    per_image_tflop = (2 * args.denoise_steps * UNET_GFLOP_PER_SAMPLE + VAE_GFLOP_PER_IMAGE) / 1e3
    out = {
        "metric": "images/sec (SD1.5 512x512, 30 steps, fp16)", "value": round(value, 4), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"SD v1.5 geometry 512x512, {args.scheduler} {args.denoise_steps} steps, CFG 7.5, batch {args.batch} per GPU "
                               f"(UNet batch {2 * args.batch}), fp16, + batched VAE decode; random-init weights, text length 77",
                   "global_batch": (world // 2 if pair else world) * B,
                   "parallelism": (f"dp{world // 2} x cfg-pair2 (uncond/text halves on paired GPUs, 2-rank RCCL all-gather of [B,4,h,w] per step)"
                                   if pair else f"dp{world} (independent prompts, no collective)"),
                   "unet_calls_per_step": calls // max(1, args.steps), "reference_quirks": True},
        "unet_it_per_s": round(calls / (unet_ms * 1e-3), 3) if unet_ms else None,
        "unet_ms_per_call": round(unet_ms / max(1, calls), 3), "vae_ms_per_batch": round(vae_ms / args.steps, 2),
        "host_enqueue_ms_per_unet_call": round(enq_ms / max(1, calls), 3),   # host time to enqueue one UNet call's launches (no waiting)
        "effective_tflops_per_gpu": round(value / world * per_image_tflop, 1),
        "mfma_roofline_frac_end_to_end": round(value / world * per_image_tflop / PEAK_TFLOPS_F16, 4),
        "model_load_s": round(load_s, 1),
        "roofline": roof, "kernels": kernel_table,
    }
    pipe.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(model_dir, args.cpu_budget)
        except Exception as e:  # the baseline is a reported side figure; never lose the GPU line over it
            out["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
