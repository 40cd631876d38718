#!/usr/bin/env python3
"""bench.py — images/sec of the txt2img hot path on MI355X (BASELINE.json metric).

A "step" is one full `StableDiffusionTxt2ImgOptions::run` over one batch: B=8 prompts, SD v1.5 geometry, 512x512,
DPM-Solver++ 30 steps, CFG 7.5 (UNet batch 16), fp16, followed by the batched VAE decode (BASELINE.json configs[1]).
Weights are random-init (no checkpoints exist offline), written once as a pyke model directory and loaded through the
product's own ONNX loader.  Text embeddings, initial latents and the image output buffer are resident in HBM before
the timed region.  With --gpus N every rank runs the same workload on its own GPU (independent prompts: no data-path
collective), value = all images / max-over-ranks time ("scaling": "weak").

Launching: under torchrun (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) each process finds
RANK / LOCAL_RANK / WORLD_SIZE in its environment.  Started plainly as `python bench.py --gpus N` (no WORLD_SIZE), this
process starts the N ranks itself -- one fresh child per GPU, before anything here has touched the GPU -- and prints rank 0's
line.  `--prompts P` switches to BASELINE configs[2]: P prompts with per-prompt seeds, sharded contiguously over the ranks and
run in micro-batches of --batch ("scaling": "strong", a step = one pass over all P prompts).
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


def cpu_baseline(model_dir, seconds_budget=60.0):
    """Oracle (torch CPU fp32 restatement of the reference's loop and of its two ORT graphs) on configs[0]: B=1, CFG 7.5 (UNet batch 2),
    Euler 20 steps, 512x512, fp32, + the VAE decode -- the REAL loop (oracle/txt2img.py), timed end to end.  If one UNet call is so
    slow that 20 of them would not fit the budget, a bounded sample is timed and extrapolated instead, and the line says so."""
    from oracle import model_dir as md, schedulers as OS, txt2img as OT, unet_ref
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    uw, ucfg = md.load_unet(model_dir)
    uw = {k: torch.from_numpy(v) for k, v in uw.items()}
    vw, vcfg = md.load_vae_decoder(model_dir)
    vw = {k: torch.from_numpy(v) for k, v in vw.items()}
    rng = np.random.default_rng(0)
    init = rng.standard_normal((1, 4, 64, 64)).astype(np.float32)
    ehs = (np.random.default_rng(1).standard_normal((2, 77, 768)) * 0.5).astype(np.float32)
    calls = []

    def unet(x, t, e):
        t0 = time.time()
        y = unet_ref.unet_forward(uw, ucfg, torch.from_numpy(x), t, torch.from_numpy(e)).numpy()
        calls.append(time.time() - t0)
        return y

    with torch.no_grad():
        t0 = time.time()
        unet(np.concatenate([init, init]) * 14.6, 999.0, ehs)     # warm-up (thread pool, allocator)
        warm = time.time() - t0
        calls.clear()
        if warm * 21 <= seconds_budget * 1.5:
            t0 = time.time()
            ref = OT.run(unet, OS.make("euler"), ehs, init, 20, 7.5)
            img = unet_ref.decode_latents(vw, vcfg, torch.from_numpy(ref["final"]))
            total = time.time() - t0
            assert torch.isfinite(img).all()
            t_unet = sum(calls) / len(calls)
            return {"value": 1.0 / total, "unit": "images/sec", "cores": cores, "kind": "port",
                    "sample": f"the whole configs[0] job, not a sample: 1 image, Euler 20 steps (20 UNet calls at batch 2, fp32, 64x64 latent, S=77) + "
                              f"VAE decode of the 512x512 image, {total:.1f} s wall; torch CPU oracle (ORT is not installed)",
                    "s_per_image": total, "unet_s_per_call": t_unet, "unet_it_per_s": 1.0 / t_unet}
        n = 0
        while n < 2 or (sum(calls) + warm < seconds_budget * 0.6 and n < 4):
            unet(np.concatenate([init, init]), 946.0, ehs)
            n += 1
        t_unet = sum(calls) / len(calls)
        t0 = time.time()
        unet_ref.decode_latents(vw, vcfg, torch.from_numpy(init * 0.18215))
        t_vae = time.time() - t0
    s_per_image = 20 * t_unet + t_vae
    return {"value": 1.0 / s_per_image, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} timed UNet forwards (batch 2, fp32, 64x64 latent, S=77) + 1 VAE decode of one 512x512 image, EXTRAPOLATED to Euler 20 "
                      f"steps (configs[0]) because the full loop would exceed the time budget on {cores} cores; torch CPU oracle (ORT is not installed)",
            "s_per_image": s_per_image, "unet_s_per_call": t_unet, "unet_it_per_s": 1.0 / t_unet}


# engine profiler class -> the HIP kernel symbol a rocprofv3 kernel trace shows for it.  Several classes can share one symbol (the
# plain and the GEGLU linears both run gemm_big_kernel<false, false, 5, 0, ..>): the roofline object is computed PER SYMBOL, so that it
# can be checked line by line against `rocprofv3 --kernel-trace --stats` (profiles/*kernel_stats*.csv).  Placeholders stand for
# template arguments that select a variant of the same kernel and are summed over: UPS (fused 2x upsample in the loader), LO (the
# split-fp16 epilogue instance: rows `..., false>` and `..., true>` of the CSV), BM / BN (tile of the 4-wave kernel), NT (16-column tiles
# per wave of the halo kernel: 5 = 160-column, 4 = 128-column block tiles), ONES / FOLD.
ROCPROF_SYMBOL = {
    "conv3x3_mfma_256x320": "gemm_big_kernel<true, UPS, 5, 0, LO>", "conv3x3_mfma_256x256": "gemm_big_kernel<true, UPS, 4, 0, LO>",
    "gemm_mfma_256x320": "gemm_big_kernel<false, false, 5, 0, LO>", "gemm_geglu_mfma_256x320": "gemm_big_kernel<false, false, 5, 0, LO>",
    "gemm_mfma_256x256": "gemm_big_kernel<false, false, 4, 0, LO>",
    "conv3x3_mfma_256x160": "gemm_midl_kernel<true, UPS, LO, 256>", "gemm_mfma_256x160": "gemm_mid_kernel<false, false, 0, LO>",
    "gemm_mfma_ldw128": "gemm_midl_kernel<false, false, LO, 128>", "conv3x3_mfma_halo": "gemm_midh_kernel<LO, UPS, NT>",
    "conv3x3_mfma": "gemm_mfma_kernel<BM, BN, true>", "gemm_mfma": "gemm_mfma_kernel<BM, BN, false>",
    "splitk_reduce": "splitk_reduce_kernel",
    "gemm_mfma_ws": "gemm_ws_kernel<5, NT, MT, EPI, R, WAVES>", "gemm_geglu_mfma_ws": "gemm_ws_kernel<5, NT, MT, EPI, R, WAVES>",
}


def rocprof_symbol(cls):
    return ROCPROF_SYMBOL.get(cls) or re.sub(r"^attn_mfma<(\d+),(\d+)>$", r"attn_mfma_kernel<\1, \2, ONES, FOLD>", cls)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as children and relay rank 0's JSON line.
    Runs before this process has initialised the GPU (children are fresh interpreters, not forks of a GPU process).
    The ranks are supervised: every rank's output goes to a log of its own, the first rank that exits non-zero ends the job (the others
    would sit in the rendezvous or in an all_reduce for ever) with that rank's exit code and the tail of its log, and the whole job has
    a time limit (SDE_BENCH_TIMEOUT seconds, default 1500)."""
    import socket
    import subprocess
    import tempfile
    logdir = tempfile.mkdtemp(prefix="sde_bench_ranks_")
    limit = float(os.environ.get("SDE_BENCH_TIMEOUT", "1500"))
    last_err = None
    for attempt in range(3):                    # the port is free when probed; another process may take it before rank 0 binds it
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs, logs = [], []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            out = open(os.path.join(logdir, f"rank{r}.out"), "w")
            err = open(os.path.join(logdir, f"rank{r}.err"), "w")
            logs.append((out, err))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out, stderr=err, text=True))
        t0 = time.time()
        failed = None
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > limit:
                failed = -1
                break
            time.sleep(0.05)
        if failed is not None:
            for p in procs:                      # end the survivors: the exact children started above, nothing by pattern
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        for out, err in logs:
            out.close(); err.close()
        tail = lambda path: "".join(open(path).readlines()[-15:])
        if failed is None:
            lines = [ln for ln in open(os.path.join(logdir, "rank0.out")).read().splitlines() if ln.startswith("{")]
            sys.stderr.write(open(os.path.join(logdir, "rank0.err")).read())
            if lines:
                print(lines[-1])
                sys.exit(0)
            print("[bench] rank 0 printed no result line", file=sys.stderr)
            sys.exit(1)
        if failed == -1:
            print(f"[bench] the {n}-rank job exceeded {limit:.0f} s and was ended; rank logs under {logdir}", file=sys.stderr)
            for r in range(n):
                sys.stderr.write(f"---- rank {r} stderr (tail)\n{tail(os.path.join(logdir, f'rank{r}.err'))}")
            sys.exit(124)
        rc = procs[failed].returncode
        err_tail = tail(os.path.join(logdir, f"rank{failed}.err"))
        if "Address already in use" in err_tail and attempt < 2:      # lost the race for the probed port: a fresh port, once more
            last_err = err_tail
            continue
        print(f"[bench] rank {failed} exited with code {rc}; the other ranks were ended.  Its output (tail; all logs under {logdir}):", file=sys.stderr)
        sys.stderr.write(tail(os.path.join(logdir, f"rank{failed}.out")))
        sys.stderr.write(err_tail)
        sys.exit(abs(rc) or 1)
    print(f"[bench] no free rendezvous port after 3 attempts: {last_err}", file=sys.stderr)
    sys.exit(1)


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
    ap.add_argument("--cpu-budget", type=float, default=60.0)
    ap.add_argument("--cfg-pair-split", action="store_true",
                    help="BASELINE config 3 variant: ranks (2p, 2p+1) share 2*batch prompts; each runs one CFG half of every UNet "
                         "call (UNet batch stays 2*batch per GPU) and they all-gather the halves over RCCL once per step")
    ap.add_argument("--prompts", type=int, default=0,
                    help="BASELINE configs[2]: this many prompts (seeds 0..P-1) sharded over the ranks, micro-batches of --batch, strong scaling")
    ap.add_argument("--rehearse", action="store_true",
                    help="launch / rendezvous / aggregation rehearsal on a box without GPUs: gloo, and a sleep stands in for the engine "
                         "(the line says so in `data`; it is not a measurement)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])      # does not return

    if os.environ.get("SDE_BENCH_DEBUG"):      # dump every thread's Python stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SDE_BENCH_DEBUG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("SDE_BENCH_TEST_FAIL_RANK") == str(rank):      # tests/test_bench_launch.py: a rank that dies before the rendezvous
        print("injected start-up failure", file=sys.stderr)
        sys.exit(7)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # SDE_BENCH_FORCE_DIST=1: initialise the process group even for ONE rank, so that a one-GPU box executes the RCCL path of this file
    # (communicator set-up, barrier, the max / sum all-reduces of job_throughput) -- the only part of the N > 1 launch it can run
    force_dist = world == 1 and os.environ.get("SDE_BENCH_FORCE_DIST") == "1" and not args.rehearse
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        # SDE_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks then share devices)
        backend = "gloo" if args.rehearse else os.environ.get("SDE_BENCH_BACKEND", "nccl")
        if not args.rehearse:
            local_rank %= max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cpu") if args.rehearse else torch.device("cuda", local_rank)
    if not args.rehearse:
        torch.cuda.set_device(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        if not args.rehearse:
            torch.cuda.synchronize()

    B, S, D, H, W = args.batch, 77, 768, 512, 512
    pair = None
    n_groups, group_idx = world, rank              # prompt shards: one per rank, or one per CFG pair
    if args.cfg_pair_split:
        from diffusers_amd.sharding import CfgPair, pair_layout
        assert dist is not None, "--cfg-pair-split needs an even number of ranks >= 2"
        group_idx, half, n_groups = pair_layout(world, rank)
        groups = [dist.new_group([2 * p, 2 * p + 1]) for p in range(n_groups)]
        pair = CfgPair(dist, groups[group_idx], half, dev)
        if not args.prompts:
            B = 2 * B                                  # prompts per PAIR; each GPU still runs a UNet batch of 2*args.batch

    load_s = 0.0
    pipe = None
    if not args.rehearse:
        model_dir = ensure_model_dir(args.model_dir, rank, world, barrier)
        from diffusers_amd import schedulers as N
        from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline, StableDiffusionTxt2ImgOptions
        t0 = time.time()
        pipe = StableDiffusionPipeline(model_dir, StableDiffusionOptions(device_id=local_rank))
        load_s = time.time() - t0

    class Acc:
        unet_ms = vae_ms = enq_ms = 0.0
        calls = 0
    acc = Acc()

    def run_batch(ehs, init, images, noise, nb):
        """one StableDiffusionTxt2ImgOptions::run over nb prompts whose inputs are resident in HBM"""
        sch = N.BY_NAME[args.scheduler].stable_diffusion_v1_optimized_default()
        ptrs = dict(ehs=ehs.data_ptr(), S=S, D=D, init_latents=init.data_ptr(), images=images.data_ptr())
        if noise is not None:
            ptrs.update(step_noise=noise.data_ptr(), step_noise_rows=noise.shape[0])
        o = StableDiffusionTxt2ImgOptions(height=H, width=W, guidance_scale=7.5, steps=args.denoise_steps, batch_size=nb, seed=0)
        r = o.run(pipe, sch, device_ptrs=ptrs, cfg_pair=pair)
        acc.unet_ms += r.unet_ms; acc.vae_ms += r.vae_ms; acc.calls += r.unet_calls; acc.enq_ms += r.host_enqueue_ms
        return r

    check = []                                          # image tensors this rank decodes (finiteness check after the timed region)
    if args.rehearse:
        units_per_step = B if not args.prompts else None

        def one_step():
            time.sleep(0.02 * (1 + rank))
            return None
        if args.prompts:
            from diffusers_amd.sharding import shard_range
            b0, e0 = shard_range(args.prompts, n_groups, group_idx)
            units_per_step = e0 - b0
    elif not args.prompts:
        # weak scaling (configs[1]): every rank runs B prompts of its own
        g = torch.Generator(device="cpu").manual_seed(1 + group_idx)          # identical inputs inside a pair
        ehs = (torch.randn(2 * B, S, D, generator=g) * 0.5).to(dev)
        init = torch.randn(B, 4, H // 8, W // 8, generator=g).to(dev)
        images = torch.empty(B, H, W, 3, device=dev, dtype=torch.float32)
        check.append(images if pair is None else (images[:(B + 1) // 2] if pair.half == 0 else images[(B + 1) // 2:]))
        units_per_step = B

        def one_step():
            return run_batch(ehs, init, images, None, B)
    else:
        # strong scaling (configs[2]): P prompts in all; this rank's (pair's) contiguous shard in micro-batches of args.batch.  Every
        # prompt brings its own embeddings and its own seed (latents from StdRng(seed), scheduler noise from StdRng(seed + 31337), as one
        # reference run per prompt would draw them), so the images do not depend on the number of GPUs or on the micro-batching.
        import ctypes as C
        from diffusers_amd import _lib
        from diffusers_amd.sharding import prompt_seed, shard_range
        b0, e0 = shard_range(args.prompts, n_groups, group_idx)
        units_per_step = e0 - b0
        h8, w8 = H // 8, W // 8
        per1 = 4 * h8 * w8
        sch0 = N.BY_NAME[args.scheduler].stable_diffusion_v1_optimized_default()
        sch0.set_timesteps(args.denoise_steps)
        nt = len(sch0.timesteps())
        needs_noise = args.scheduler in ("euler_a", "ddpm")

        def draws(seed, skip, n):
            out = np.empty(n, np.float32)
            _lib.check(_lib.lib().sde_rng_standard_normal(C.c_uint64(seed), C.c_uint64(skip), out.ctypes.data, n))
            return out

        batches = []
        for m0 in range(b0, e0, args.batch):
            idx = list(range(m0, min(e0, m0 + args.batch)))
            nb = len(idx)
            un, tx, ini, nz = [], [], [], []
            for pidx in idx:
                g = torch.Generator(device="cpu").manual_seed(10_000 + pidx)
                e2 = torch.randn(2, S, D, generator=g) * 0.5
                un.append(e2[0]); tx.append(e2[1])
                seed = prompt_seed(0, pidx)
                ini.append(torch.from_numpy(draws(seed, 0, per1)).view(4, h8, w8))
                if needs_noise:
                    nz.append(torch.from_numpy(draws((seed + 31337) & 0xFFFFFFFFFFFFFFFF, 0, nt * per1)).view(nt, 4, h8, w8))
            ehs_b = torch.stack(un + tx).to(dev)
            init_b = torch.stack(ini).to(dev)
            noise_b = torch.stack(nz, dim=1).contiguous().to(dev) if needs_noise else None      # [nt, nb, 4, h, w]
            img_b = torch.empty(nb, H, W, 3, device=dev, dtype=torch.float32)
            batches.append((ehs_b, init_b, img_b, noise_b, nb))
            check.append(img_b if pair is None else (img_b[:(nb + 1) // 2] if pair.half == 0 else img_b[(nb + 1) // 2:]))

        def one_step():
            r = None
            for bt in batches:
                r = run_batch(*bt)
            return r

    for _ in range(args.warmup):
        one_step()
    acc.unet_ms = acc.vae_ms = acc.enq_ms = 0.0; acc.calls = 0
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync(); barrier(); sync()
    elapsed = time.perf_counter() - t0
    unet_ms, vae_ms, enq_ms, calls = acc.unet_ms, acc.vae_ms, acc.enq_ms, acc.calls
    units = float(units_per_step * args.steps) if (pair is None or pair.half == 0) else 0.0   # a pair's images are counted once
    if dist is not None:
        on = dev if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=on, dtype=torch.float64)
        u = torch.tensor([units], device=on, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        elapsed, units = float(t.item()), float(u.item())
    for im in check:
        assert torch.isfinite(im).all(), "non-finite pixels in the decoded images"

    # per-kernel timing: same workload once more with HIP events around every launch (on the engine's stream)
    roof = None
    kernel_table = []
    if pair is not None and rank != 0 and not args.rehearse:
        one_step()                      # the pair's peer (and every other pair, to keep ranks aligned) takes part in the profiled pass
    if rank == 0 and not args.rehearse:
        pipe.profile_reset(); pipe.profile(True)
        rp = one_step()
        stats = pipe.profile_get()
        pipe.profile(False)
        tot = sum(s["total_ms"] for s in stats) or 1.0
        for s in sorted(stats, key=lambda s: -s["total_ms"]):
            avg = s["total_ms"] / max(1, s["launches"])
            tf = s["flops"] / (s["total_ms"] * 1e-3) / 1e12 if s["flops"] and s["total_ms"] else None
            gb = s["bytes"] / (s["total_ms"] * 1e-3) / 1e9 if s["bytes"] and s["total_ms"] else None
            # each class against BOTH roofs (algorithmic flops and bytes): `bound` is the roof it sits closer to
            fm, fh = (tf / PEAK_TFLOPS_F16 if tf else 0.0), (gb / PEAK_HBM_GBS if gb else 0.0)
            kernel_table.append(dict(name=s["name"], launches=s["launches"], total_ms=round(s["total_ms"], 3), avg_us=round(avg * 1e3, 2),
                                     share=round(s["total_ms"] / tot, 4), tflops=round(tf, 1) if tf else None, gbs=round(gb, 1) if gb else None,
                                     bound="mfma" if fm >= fh else "hbm", roof_frac=round(max(fm, fh), 3)))
        # dominant kernel = the rocprof symbol with the largest summed time (classes that share a symbol are merged)
        by_sym = {}
        for st in stats:
            g = by_sym.setdefault(rocprof_symbol(st["name"]), dict(classes=[], total_ms=0.0, flops=0.0, bytes=0.0, launches=0))
            g["classes"].append(st["name"]); g["total_ms"] += st["total_ms"]; g["flops"] += st["flops"]; g["bytes"] += st["bytes"]
            g["launches"] += st["launches"]
        sym, dom = max(by_sym.items(), key=lambda kv: kv[1]["total_ms"])
        # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this process, so the figure comes from the
        # committed rocprofv3 --pmc passes over this same workload, 30 denoise steps (profiles/r3_pmc_traffic.json, FETCH_SIZE / WRITE_SIZE in separate runs)
        traffic = None
        try:
            pmc_path = next(p for p in (os.path.join(ROOT, "profiles", f) for f in ("r3_pmc_traffic.json", "r2_pmc_traffic.json")) if os.path.exists(p))
            pmc = json.load(open(pmc_path))["symbols"]
            v = pmc.get(sym)
            if v:
                traffic = round((v["fetch_MB_per_launch_corrected_x2"] + v["write_MB_per_launch"]) * 1e6)
        except Exception:
            traffic = None
        common = {"kernel": "+".join(dom["classes"]), "rocprof_symbol": sym, "traffic": traffic, "launches": dom["launches"],
                  "avg_launch_us": round(dom["total_ms"] / dom["launches"] * 1e3, 2), "share_of_gpu_time": round(dom["total_ms"] / tot, 4),
                  "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]), "profiled_step_ms": round(rp.total_ms, 2)}
        if dom["flops"] > 0:
            ach = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS_F16, 4),
                    "flop_per_launch": dom["flops"] / dom["launches"],
                    # the same launches against the other roof: a class of short-K linears sits near the ridge (2500 / 8 = 312 flop/B)
                    "flop_per_byte": round(dom["flops"] / dom["bytes"], 1) if dom["bytes"] else None,
                    "hbm_frac": round(dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if dom["bytes"] else None, **common}
        else:
            ach = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), **common}

    value = units / elapsed
    strong = args.prompts > 0
    if strong:
        workload = (f"SD v1.5 geometry 512x512, {args.scheduler} {args.denoise_steps} steps, CFG 7.5, {args.prompts} prompts (seeds 0..{args.prompts - 1}) "
                    f"sharded over {n_groups} {'GPU pairs' if pair else 'GPUs'} in micro-batches of {args.batch}, fp16, + VAE decode; random-init weights, text length 77")
    else:
        workload = (f"SD v1.5 geometry 512x512, {args.scheduler} {args.denoise_steps} steps, CFG 7.5, batch {args.batch} per GPU "
                    f"(UNet batch {2 * args.batch}), fp16, + batched VAE decode; random-init weights, text length 77")
    per_image_tflop = (2 * args.denoise_steps * UNET_GFLOP_PER_SAMPLE + VAE_GFLOP_PER_IMAGE) / 1e3
    out = {
        "metric": f"images/sec (SD1.5 512x512, {args.denoise_steps} steps, fp16)", "value": round(value, 4), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f16",
        "data": "rehearsal: a sleep stands in for the engine, NOT a measurement" if args.rehearse else "synthetic",
        "config": {"workload": workload,
                   "global_batch": args.prompts if strong else n_groups * B,
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
    if pipe is not None:
        pipe.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.rehearse:
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
