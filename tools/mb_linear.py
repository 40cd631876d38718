#!/usr/bin/env python3
"""Short-K linears of the 64x64 / 32x32 UNet levels at UNet batch 16, per kernel (HIP events inside the engine): the tile kernels
(MB_KERNELS: 2 = 256x320, 3 = 256x160) against the W-stationary streaming kernel (6, gemm_ws.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from diffusers_amd import _lib, modelgen
from diffusers_amd.pipeline import StableDiffusionPipeline

SHAPES = [  # M, K, N, kind (std | lo = split-fp16 residual and result | geglu)
    (65536, 320, 320, "std"), (65536, 320, 960, "std"), (65536, 640, 320, "lo"), (65536, 320, 2560, "geglu"),
    (16384, 640, 640, "std"), (16384, 640, 1920, "std"), (16384, 640, 5120, "geglu"), (16384, 640, 640, "lo"), (65536, 640, 320, "std"),
    (4096, 1280, 3840, "std"), (4096, 1280, 1280, "std"), (16384, 2560, 640, "std"), (4096, 5120, 1280, "std"), (65536, 1280, 320, "std"),   # 9 .. 13
]


def main():
    d = "/tmp/mb_tiny"
    if not os.path.exists(d + "/unet.onnx"):
        modelgen.write_model_dir(d, modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False),
                                                         cross_attention_dim=64, num_heads=4),
                                 modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1), seed=1)
    p = StableDiffusionPipeline(d)
    L = _lib.lib()
    rng = np.random.default_rng(0)
    reps = int(os.environ.get("MB_REPS", "4"))
    kerns = [int(k) for k in os.environ.get("MB_KERNELS", "2,3,6").split(",")]
    sel = os.environ.get("MB_SHAPES")      # e.g. "0,1": indices into SHAPES
    shapes = [SHAPES[int(i)] for i in sel.split(",")] if sel else SHAPES
    for (M, K, N, kind) in shapes:
        x = rng.standard_normal((M, K), dtype=np.float32)
        w = (rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).astype(np.float32)
        res = rng.standard_normal((M, N), dtype=np.float32) if kind == "lo" else None
        out = np.empty((M, N // 2 if kind == "geglu" else N), np.float32)
        for kern in kerns:
            def call():
                if kind == "geglu":
                    return L.sde_test_geglu(p._h, kern, x.ctypes.data, M, K, w.ctypes.data, None, N, 0, out.ctypes.data, out.size)
                if kind == "lo":
                    return L.sde_test_conv_res(p._h, kern, x.ctypes.data, 1, M // 64, 64, K, w.ctypes.data, None, N, 1, res.ctypes.data, 1, out.ctypes.data, out.size)
                return L.sde_test_conv(p._h, kern, x.ctypes.data, 1, M // 64, 64, K, w.ctypes.data, None, N, 1, 1, 0, out.ctypes.data, out.size)
            try:
                _lib.check(call())
            except Exception as e:
                print(f"{kind:5s} M={M} K={K} N={N} kern={kern}: {e}", flush=True)
                continue
            p.profile_reset(); p.profile(True)
            for _ in range(reps):
                _lib.check(call())
            st = [s for s in p.profile_get() if "mfma" in s["name"]]
            p.profile(False)
            for s in st:
                us = s["total_ms"] / s["launches"] * 1e3
                print(f"{kind:5s} M={M:6d} K={K:4d} N={N:5d} kern={kern} {s['name']:26s} {us:8.1f} us  {s['flops']/s['launches']/us/1e6:7.1f} TFLOP/s  "
                      f"{s['bytes']/s['launches']/us/1e3:7.1f} GB/s", flush=True)
    p.close()


if __name__ == "__main__":
    main()
