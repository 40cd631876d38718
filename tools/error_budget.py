#!/usr/bin/env python3
"""Where does the fp16 engine's error against the fp32 oracle come from?  (CPU tool; test infrastructure, not product.)

Re-runs the oracle's UNet (oracle/unet_ref.py arithmetic) with a rounding to fp16 inserted at the points where the ENGINE stores
an fp16 tensor, one class of points at a time, and reports the relative L2 error of each UNet half and of the CFG-combined
prediction u + g (t - u) against the unrounded run.  Classes:

  in      sample / text embeddings packed to fp16
  gn      GroupNorm(+SiLU) outputs            ln      LayerNorm outputs
  conv    conv / linear outputs that are NOT a residual sum (conv1, q/k/v, proj_in, branch outputs y0..y3, ff1 via geglu)
  resid   outputs of epilogues that add a residual (resnet conv2 + shortcut, proj_out + x) and the block-final sum
  stream  (old engine) the transformer's residual stream rounded after every sub-layer
  attn_p  softmax probabilities (the P operand of the PV MFMA)        attn_o  attention outputs
  geglu   the a * gelu(g) tensor

usage: python tools/error_budget.py [--model-dir DIR] [--S 77] [--t 999] [--classes all|name,name...]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F

from oracle import model_dir as md, unet_ref as U

ON = set()


def Q(x, cls):
    """cls or cls@<latent width> (NCHW tensors: width; token tensors [B, HW, C]: sqrt(HW))"""
    if cls in ON:
        return x.half().float()
    wd = x.shape[-1] if x.dim() == 4 else int(round(x.shape[1] ** 0.5))
    return x.half().float() if f"{cls}@{wd}" in ON else x


def resnet(x, emb, w, p, groups, eps):
    h = Q(F.silu(U.group_norm(x, w, p + ".norm1", groups, eps)), "gn")
    h = U.conv(h, w, p + ".conv1")
    if emb is not None and (p + ".time_emb_proj.weight") in w:
        h = h + U.linear(F.silu(emb), w, p + ".time_emb_proj")[:, :, None, None]
    h = Q(Q(h, "conv"), "c_conv1")
    h = Q(F.silu(U.group_norm(h, w, p + ".norm2", groups, eps)), "gn")
    h = U.conv(h, w, p + ".conv2")
    if (p + ".conv_shortcut.weight") in w:
        x = Q(Q(U.conv(Q(x, "op_short"), w, p + ".conv_shortcut", padding=0), "conv"), "c_short")
    return Q(Q(x + h, "resid"), "r_res")


def attention(x, ctx, w, p, heads):
    q = Q(Q(U.linear(x, w, p + ".to_q"), "conv"), "c_qkv")
    k = Q(Q(U.linear(ctx, w, p + ".to_k"), "conv"), "c_qkv")
    v = Q(Q(U.linear(ctx, w, p + ".to_v"), "conv"), "c_qkv")
    B, S, C = q.shape
    d = C // heads
    split = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    att = Q(torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1), "attn_p")
    o = Q((att @ v).transpose(1, 2).reshape(B, S, C), "attn_o")
    return Q(Q(U.linear(o, w, p + ".to_out.0"), "conv"), "c_y")


def transformer(x, ctx, w, p, heads, groups, linear_proj, gelu="none"):
    B, C, H, W = x.shape
    r = x
    h = Q(U.group_norm(x, w, p + ".norm", groups, 1e-6), "gn")
    if not linear_proj:
        h = Q(Q(U.conv(h, w, p + ".proj_in", padding=0), "conv"), "c_y")
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    else:
        h = Q(Q(U.linear(h.permute(0, 2, 3, 1).reshape(B, H * W, C), w, p + ".proj_in"), "conv"), "c_y")
    b = p + ".transformer_blocks.0"
    ln = lambda t, n: Q(F.layer_norm(t, (C,), w[f"{b}.{n}.weight"], w[f"{b}.{n}.bias"], 1e-5), "ln")
    n1 = ln(h, "norm1")
    h = Q(h + attention(n1, n1, w, b + ".attn1", heads), "stream")
    h = Q(h + attention(ln(h, "norm2"), ctx, w, b + ".attn2", heads), "stream")
    y = U.linear(ln(h, "norm3"), w, b + ".ff.net.0.proj")
    a, g = y.chunk(2, dim=-1)
    h = Q(Q(h + Q(Q(U.linear(Q(a * F.gelu(g), "geglu"), w, b + ".ff.net.2"), "conv"), "c_y"), "resid"), "r_final")
    if not linear_proj:
        h = U.conv(h.reshape(B, H, W, C).permute(0, 3, 1, 2), w, p + ".proj_out", padding=0)
    else:
        h = U.linear(h, w, p + ".proj_out").reshape(B, H, W, C).permute(0, 3, 1, 2)
    return Q(Q(h + r, "resid"), "r_proj")


# ---- "v2": every tensor that is not an MFMA operand stays fp32 (GEMM epilogues store fp32, norms read fp32); only operands are rounded
def resnet_v2(x, emb, w, p, groups, eps):
    h = Q(F.silu(U.group_norm(x, w, p + ".norm1", groups, eps)), "v2")
    h = U.conv(h, w, p + ".conv1")
    if emb is not None and (p + ".time_emb_proj.weight") in w:
        h = h + U.linear(F.silu(emb), w, p + ".time_emb_proj")[:, :, None, None]
    h = Q(F.silu(U.group_norm(h, w, p + ".norm2", groups, eps)), "v2")
    h = U.conv(h, w, p + ".conv2")
    if (p + ".conv_shortcut.weight") in w:
        x = U.conv(Q(x, "v2"), w, p + ".conv_shortcut", padding=0)
    return x + h


def attention_v2(x, ctx, w, p, heads):
    q = Q(U.linear(x, w, p + ".to_q"), "v2")
    k = Q(U.linear(ctx, w, p + ".to_k"), "v2")
    v = Q(U.linear(ctx, w, p + ".to_v"), "v2")
    B, S, C = q.shape
    d = C // heads
    split = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    att = Q(torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1), "v2")
    o = Q((att @ v).transpose(1, 2).reshape(B, S, C), "v2")
    return U.linear(o, w, p + ".to_out.0")


def transformer_v2(x, ctx, w, p, heads, groups, linear_proj, gelu="none"):
    B, C, H, W = x.shape
    r = x
    h = Q(U.group_norm(x, w, p + ".norm", groups, 1e-6), "v2")
    h = U.conv(h, w, p + ".proj_in", padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    b = p + ".transformer_blocks.0"
    ln = lambda t, n: Q(F.layer_norm(t, (C,), w[f"{b}.{n}.weight"], w[f"{b}.{n}.bias"], 1e-5), "v2")
    n1 = ln(h, "norm1")
    h = h + attention_v2(n1, n1, w, b + ".attn1", heads)
    h = h + attention_v2(ln(h, "norm2"), ctx, w, b + ".attn2", heads)
    y = U.linear(ln(h, "norm3"), w, b + ".ff.net.0.proj")
    a, g = y.chunk(2, dim=-1)
    h = Q(h + U.linear(Q(a * F.gelu(g), "v2"), w, b + ".ff.net.2"), "v2")
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return U.conv(h, w, p + ".proj_out", padding=0) + r


def forward_v2(w, cfg, sample, t, ehs):
    saved = (U.resnet, U.transformer, U.conv)
    orig_conv = U.conv

    def conv(x_, w_, p_, stride=1, padding=1):
        if p_ == "conv_in" or "samplers" in p_:
            x_ = Q(x_, "v2")
        return orig_conv(x_, w_, p_, stride, padding)
    U.resnet, U.transformer, U.conv = resnet_v2, transformer_v2, conv
    try:
        return U.unet_forward(w, cfg, sample, t, Q(ehs, "v2"))
    finally:
        U.resnet, U.transformer, U.conv = saved


def forward(w, cfg, sample, t, ehs):
    """oracle.unet_ref.unet_forward with the rounding points in (the oracle module's own blocks are swapped for the ones above)."""
    saved = (U.resnet, U.transformer)
    U.resnet, U.transformer = resnet, transformer
    try:
        x, e = Q(sample, "in"), Q(ehs, "in")
        orig_conv = U.conv

        def conv(x_, w_, p_, stride=1, padding=1):
            if "samplers" in p_:
                x_ = Q(x_, "op_samp")
            y = orig_conv(x_, w_, p_, stride, padding)
            return Q(Q(y, "conv"), "c_samp") if p_ in ("conv_in",) or "samplers" in p_ else y
        U.conv = conv
        try:
            return U.unet_forward(w, cfg, x, t, e)
        finally:
            U.conv = orig_conv
    finally:
        U.resnet, U.transformer = saved


def rel(a, b):
    return float(torch.linalg.norm((a - b).double()) / torch.linalg.norm(b.double()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir", default=os.environ.get("SDE_BENCH_MODEL_DIR", "/tmp/sde_bench_sd15_f16"))
    ap.add_argument("--S", type=int, default=77)
    ap.add_argument("--t", type=float, default=999.0)
    ap.add_argument("--g", type=float, default=7.5)
    ap.add_argument("--scale", type=float, default=1.0, help="std of the sample fed to the UNet")
    ap.add_argument("--classes", default="all")
    ap.add_argument("--sd21", action="store_true", help="SD v2.1 geometry at 96x96 (model dir /tmp/sde_cfg_sd21_f16)")
    args = ap.parse_args()
    if args.sd21:
        args.model_dir = os.environ.get("SDE_SD21_MODEL_DIR", "/tmp/sde_cfg_sd21_f16")
    if not os.path.exists(os.path.join(args.model_dir, ".complete")):
        from diffusers_amd import modelgen
        modelgen.write_model_dir(args.model_dir, modelgen.sd21_unet_config() if args.sd21 else modelgen.unet_config(), modelgen.vae_config(), seed=42,
                                 dtype=np.float16)
        open(os.path.join(args.model_dir, ".complete"), "w").write("ok")
    uw, cfg = md.load_unet(args.model_dir)
    uw = {k: torch.from_numpy(v) for k, v in uw.items()}
    rng = np.random.default_rng(16000 + args.S)
    hw = 96 if args.sd21 else 64
    lat = torch.from_numpy((rng.standard_normal((1, 4, hw, hw)) * args.scale).astype(np.float32))
    x = torch.cat([lat, lat], 0)
    ehs = torch.from_numpy((rng.standard_normal((2, args.S, 1024 if args.sd21 else 768)) * 0.5).astype(np.float32))
    all_cls = ["in", "gn", "ln", "conv", "resid", "stream", "attn_p", "attn_o", "geglu"]
    if args.classes == "all":
        runs = [[c] for c in all_cls] + [[c for c in all_cls if c != "stream"], all_cls]
    elif args.classes == "fine":
        runs = [[c] for c in ["c_conv1", "c_short", "c_qkv", "c_y", "c_samp", "r_res", "r_final", "r_proj"]]
    else:
        CUR = "in,gn,ln,c_conv1,c_qkv,c_y,r_final,attn_p,attn_o,geglu,op_short,op_samp"   # the engine with its split-fp16 trunk
        runs = [[x for x in c.replace("CUR", CUR).split(",") if not x.startswith("-")] for c in args.classes.split(";")]
        runs = [[x for x in r if ("-" + x) not in c.split(",")] for r, c in zip(runs, args.classes.split(";"))]
    with torch.no_grad():
        ON.clear()
        t0 = time.time()
        ref = forward(uw, cfg, x, args.t, ehs)
        print(f"fp32 run {time.time() - t0:.1f}s; |t-u|/|u| = {rel(ref[1], ref[0]):.3f}", flush=True)
        refc = ref[0] + args.g * (ref[1] - ref[0])
        print(f"{'rounded classes':40s} {'uncond':>9s} {'text':>9s} {'cfg g=' + str(args.g):>11s} {'t-u':>9s}")
        for cls in runs:
            ON.clear(); ON.update(cls)
            out = forward_v2(uw, cfg, x, args.t, ehs) if cls == ["v2"] else forward(uw, cfg, x, args.t, ehs)
            c = out[0] + args.g * (out[1] - out[0])
            print(f"{'+'.join(cls):40s} {rel(out[0], ref[0]):9.2e} {rel(out[1], ref[1]):9.2e} {rel(c, refc):11.2e} "
                  f"{rel(out[1] - out[0], ref[1] - ref[0]):9.2e}", flush=True)


if __name__ == "__main__":
    main()
