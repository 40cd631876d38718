#!/usr/bin/env python3
"""Randomised parity sweep of the contraction kernels through the C ABI test hooks (sde_test_conv / sde_test_attention) against a
torch fp32 reference on fp16-rounded operands.  Shapes are drawn from the ranges the UNet / VAE produce (including odd spatial
sizes, ragged tiles, cropped upsampling, every kernel family via the `naive` selector).  Prints the worst cases; exit code 1 on a
miss.  usage: python tools/fuzz_kernels.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusers_amd import _lib, modelgen
from diffusers_amd.pipeline import StableDiffusionPipeline


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    d = "/tmp/mb_tiny"
    if not os.path.exists(d + "/unet.onnx"):
        modelgen.write_model_dir(d, modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False),
                                                         cross_attention_dim=64, num_heads=4),
                                 modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1), seed=1)
    pipe = StableDiffusionPipeline(d)
    L = _lib.lib()
    rng = np.random.default_rng(seed)
    worst, bad = [], 0
    for it in range(n_cases):
        if it % 3 != 2:
            k = int(rng.choice([1, 3, 3]))
            cin = int(rng.choice([64, 128, 192, 320, 640, 24]))
            cout = int(rng.choice([160, 320, 128, 256, 480, 640, 4, 40]))
            H, W = int(rng.integers(1, 34)), int(rng.integers(1, 34))
            if rng.random() < 0.4:   # whole 16 x 16 patches (8 x 8 source patches under upsampling): the halo kernel's domain
                H, W = int(rng.choice([8, 16, 32, 48])), int(rng.choice([8, 16, 32, 48]))
            NB = int(rng.integers(1, 4))
            stride = int(rng.choice([1, 1, 2])) if k == 3 else 1
            ups = int(rng.choice([0, 0, 1, 2])) if (k == 3 and stride == 1) else 0
            if ups == 2 and (H < 2 or W < 2):
                ups = 1
            kern = int(rng.choice([0, 0, 2, 3, 4, 5, 5]))   # 4 / 5: the 128x160 linear and halo 3x3 kernels where they support the case
            x = rng.standard_normal((NB, H, W, cin)).astype(np.float32)
            w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
            b = rng.standard_normal(cout).astype(np.float32)
            Hv, Wv = (2 * H - (ups == 2), 2 * W - (ups == 2)) if ups else (H, W)
            Ho, Wo = ((Hv - 1) // stride + 1, (Wv - 1) // stride + 1) if k == 3 else (Hv, Wv)
            out = np.empty((NB, Ho, Wo, cout), np.float32)
            _lib.check(L.sde_test_conv(pipe._h, kern, x.ctypes.data, NB, H, W, cin, w.ctypes.data, b.ctypes.data, cout, k, stride, ups,
                                       out.ctypes.data, out.size))
            xt = torch.from_numpy(x).half().float().permute(0, 3, 1, 2)
            if ups:
                xt = F.interpolate(xt, size=(Hv, Wv), mode="nearest")
            ref = F.conv2d(xt, torch.from_numpy(w).half().float(), torch.from_numpy(b), stride=stride if k == 3 else 1,
                           padding=1 if k == 3 else 0).permute(0, 2, 3, 1).numpy()
            err, tag = rel_l2(out, ref), f"conv kern={kern} NB={NB} {H}x{W} {cin}->{cout} k{k} s{stride} u{ups}"
        else:
            dd = int(rng.choice([40, 64, 80, 160, 16, 48, 12]))
            heads = int(rng.integers(1, 9))
            Sq = int(rng.integers(1, 700))
            Skv = Sq if rng.random() < 0.5 else int(rng.choice([77, 152, 227, int(rng.integers(1, 300))]))
            NB = int(rng.integers(1, 3))
            kern = int(rng.choice([0, 0, 1]))
            q = rng.standard_normal((NB, Sq, heads * dd)).astype(np.float32)
            kk = rng.standard_normal((NB, Skv, heads * dd)).astype(np.float32)
            v = rng.standard_normal((NB, Skv, heads * dd)).astype(np.float32)
            out = np.empty_like(q)
            _lib.check(L.sde_test_attention(pipe._h, kern, q.ctypes.data, kk.ctypes.data, v.ctypes.data, NB, heads, Sq, Skv, dd, out.ctypes.data, out.size))
            sp = lambda t, S: torch.from_numpy(t).half().float().reshape(NB, S, heads, dd).permute(0, 2, 1, 3)
            ref = F.scaled_dot_product_attention(sp(q, Sq), sp(kk, Skv), sp(v, Skv)).permute(0, 2, 1, 3).reshape(NB, Sq, heads * dd).numpy()
            err, tag = rel_l2(out, ref), f"attn kern={kern} NB={NB} heads={heads} Sq={Sq} Skv={Skv} d={dd}"
        worst.append((err, tag))
        if not np.isfinite(err) or err > 3e-3:
            bad += 1
            print("MISS", err, tag, flush=True)
    worst.sort(reverse=True)
    for e, t in worst[:8]:
        print(f"{e:.2e}  {t}")
    print(f"{n_cases} cases, {bad} misses")
    pipe.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
