#!/usr/bin/env python3
"""Per-shape kernel timings (HIP events inside the engine) for the SD1.5 UNet's contraction shapes at UNet batch 16."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from diffusers_amd import _lib, modelgen
from diffusers_amd.pipeline import StableDiffusionPipeline

NB = int(os.environ.get("NB", "16"))
KERN = int(os.environ.get("MB_KERNEL", "0"))   # 0 heuristic, 2 / 3 force the 256x320 / 256x160 kernel (with SDE_SPLITK=n); 7 / 8: the fp32 engine's MFMA / plain kernels
CONVS = [  # H, Cin, Cout, k, stride, ups
    (64, 320, 320, 3, 1, 0), (64, 640, 320, 3, 1, 0), (32, 640, 640, 3, 1, 0), (32, 1280, 640, 3, 1, 0), (16, 1280, 1280, 3, 1, 0),
    (16, 2560, 1280, 3, 1, 0), (8, 1280, 1280, 3, 1, 0), (8, 2560, 1280, 3, 1, 0), (32, 320, 320, 3, 1, 1), (64, 320, 320, 3, 2, 0),
    (64, 320, 320, 1, 1, 0), (64, 320, 960, 1, 1, 0), (64, 1280, 320, 1, 1, 0), (32, 640, 640, 1, 1, 0), (32, 2560, 640, 1, 1, 0),
    (16, 1280, 1280, 1, 1, 0), (16, 2560, 1280, 1, 1, 0), (16, 5120, 1280, 1, 1, 0), (64, 320, 2560, 1, 1, 0),
    (64, 64, 320, 1, 1, 0), (64, 64, 2560, 1, 1, 0),   # one K-tile: prologue + epilogue cost of the 256x320 kernel (1 and 8 rounds of blocks)
]
VAE = [  # NB=8 shapes of the SD VAE decoder (MB_ONLY=vae NB=8): H, Cin, Cout, k, stride, ups
    (64, 512, 512, 3, 1, 0), (128, 512, 512, 3, 1, 0), (64, 512, 512, 3, 1, 1), (256, 512, 256, 3, 1, 0), (256, 256, 256, 3, 1, 0),
    (128, 512, 512, 3, 1, 1), (512, 256, 128, 3, 1, 0), (512, 128, 128, 3, 1, 0), (256, 256, 256, 3, 1, 1),
]
ATTN = [(8, 4096, 4096, 40), (8, 1024, 1024, 80), (8, 256, 256, 160), (8, 4096, 77, 40), (8, 1024, 77, 80)]


def main():
    d = "/tmp/mb_tiny"
    if not os.path.exists(d + "/unet.onnx"):
        modelgen.write_model_dir(d, modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False),
                                                         cross_attention_dim=64, num_heads=4),
                                 modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1), seed=1)
    p = StableDiffusionPipeline(d)
    L = _lib.lib()
    rng = np.random.default_rng(0)
    reps = 4
    only = os.environ.get("MB_ONLY", "")
    convs = VAE if only == "vae" else [c for c in CONVS if not only or only == "convs" or (only == "small" and NB * c[0] * c[0] <= 4096) or (only == "big" and NB * c[0] * c[0] > 4096)
             or (only == "c64" and c[0] == 64 and c[3] == 3 and c[4] == 1) or (only == "k64" and c[1] == 64) or (only == "wide" and c[3] == 1 and c[2] >= 960) or (only == "lin16" and c[0] == 16 and c[3] == 1) or (only == "lin64" and c[0] == 64 and c[3] == 1 and c[1] >= 320) or (only == "c32" and c[0] == 32 and c[3] == 3 and c[5] == 0)]
    for (H, Cin, Cout, k, stride, ups) in convs:
        x = rng.standard_normal((NB, H, H, Cin), dtype=np.float32)
        w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.sqrt(Cin * k * k)).astype(np.float32)
        Hv = 2 * H if ups else H
        Ho = (Hv - 1) // stride + 1 if k == 3 else Hv
        out = np.empty((NB, Ho, Ho, Cout), np.float32)
        _lib.check(L.sde_test_conv(p._h, KERN, x.ctypes.data, NB, H, H, Cin, w.ctypes.data, None, Cout, k, stride, ups, out.ctypes.data, out.size))
        p.profile_reset(); p.profile(True)
        for _ in range(reps):
            _lib.check(L.sde_test_conv(p._h, KERN, x.ctypes.data, NB, H, H, Cin, w.ctypes.data, None, Cout, k, stride, ups, out.ctypes.data, out.size))
        st = [s for s in p.profile_get() if "mfma" in s["name"] or s["name"] in ("conv3x3_f32", "gemm_f32")]
        p.profile(False)
        for s in st:
            us = s["total_ms"] / s["launches"] * 1e3
            print(f"conv k{k} s{stride} u{ups} {H:3d}^2 {Cin:5d}->{Cout:5d}  M={NB*Ho*Ho:6d}  {us:9.1f} us  {s['flops']/s['launches']/us/1e6:7.1f} TFLOP/s", flush=True)
    for (heads, Sq, Skv, dd) in (ATTN if only in ("", "attn") else []):
        q = rng.standard_normal((NB, Sq, heads * dd), dtype=np.float32)
        kk = rng.standard_normal((NB, Skv, heads * dd), dtype=np.float32)
        v = rng.standard_normal((NB, Skv, heads * dd), dtype=np.float32)
        out = np.empty_like(q)
        _lib.check(L.sde_test_attention(p._h, KERN if KERN >= 7 else 0, q.ctypes.data, kk.ctypes.data, v.ctypes.data, NB, heads, Sq, Skv, dd, out.ctypes.data, out.size))
        p.profile_reset(); p.profile(True)
        for _ in range(reps):
            _lib.check(L.sde_test_attention(p._h, KERN if KERN >= 7 else 0, q.ctypes.data, kk.ctypes.data, v.ctypes.data, NB, heads, Sq, Skv, dd, out.ctypes.data, out.size))
        st = [s for s in p.profile_get() if "attn" in s["name"]]
        p.profile(False)
        for s in st:
            us = s["total_ms"] / s["launches"] * 1e3
            print(f"attn h{heads} Sq={Sq} Skv={Skv} d={dd}  {us:9.1f} us  {s['flops']/s['launches']/us/1e6:7.1f} TFLOP/s", flush=True)
    p.close()


if __name__ == "__main__":
    main()
