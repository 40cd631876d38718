"""ORACLE / TEST INFRASTRUCTURE — not part of the shipped product path.

torch-CPU fp32 functional restatement of the two ONNX graphs the reference executes through
ONNX Runtime on its hot path:

  * ``session.unet.run([sample, timestep, encoder_hidden_states])``
    (/root/reference/src/pipelines/stable_diffusion/impl_txt2img.rs:283), i.e. the
    ``UNet2DConditionModelIOWrapper`` exported by /root/reference/scripts/sd2pyke.py:113-125,284-301
  * ``vae_decoder.run([latent_sample])`` (impl_main.rs:403), i.e. ``AutoencoderKLIOWrapper.decode``
    (sd2pyke.py:188-192,348-357)

The layer arithmetic lives in a third-party dependency that is NOT under /root/reference:
**diffusers==0.14.0** (/root/reference/requirements.txt:3), exported with torch 1.13 at opset 15.
Its published architecture is restated here.  The VAE half is pinned
against the reference's own fixture graph tests/stable-diffusion/vae_decoder.onnx through
oracle/onnx_interp.py (tests/test_oracle_vae_fixture.py).  The UNet half is **parity unpinned**:
the reference's tests/stable-diffusion/unet.onnx is a missing LFS blob and no ORT exists here.

Weights are a dict {pytorch_parameter_name: torch.Tensor fp32}.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

T = torch.Tensor


def default_unet_config(**over) -> dict:
    cfg = dict(
        in_channels=4, out_channels=4,
        block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
        down_attn=(True, True, True, False),
        cross_attention_dim=768,
        num_heads=8,               # diffusers 0.14 "attention_head_dim" (int or per-block list) = HEAD COUNT
        use_linear_projection=False,
        norm_num_groups=32, norm_eps=1e-5,
    )
    cfg.update(over)
    return cfg


def sd21_unet_config(**over) -> dict:
    return default_unet_config(cross_attention_dim=1024, num_heads=(5, 10, 20, 20), use_linear_projection=True, **over)


def heads_for(cfg, level: int) -> int:
    h = cfg["num_heads"]
    return h if isinstance(h, int) else h[level]


# --------------------------------------------------------------------------- building blocks


def timestep_embedding(t: T, dim: int) -> T:
    """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def group_norm(x: T, w: Dict[str, T], p: str, groups: int, eps: float) -> T:
    return F.group_norm(x, groups, w[p + ".weight"], w[p + ".bias"], eps)


def conv(x: T, w: Dict[str, T], p: str, stride=1, padding=1) -> T:
    return F.conv2d(x, w[p + ".weight"], w.get(p + ".bias"), stride=stride, padding=padding)


def linear(x: T, w: Dict[str, T], p: str) -> T:
    return F.linear(x, w[p + ".weight"], w.get(p + ".bias"))


def resnet(x: T, emb: T, w, p: str, groups: int, eps: float) -> T:
    h = F.silu(group_norm(x, w, p + ".norm1", groups, eps))
    h = conv(h, w, p + ".conv1")
    if emb is not None and (p + ".time_emb_proj.weight") in w:
        h = h + linear(F.silu(emb), w, p + ".time_emb_proj")[:, :, None, None]
    h = F.silu(group_norm(h, w, p + ".norm2", groups, eps))
    h = conv(h, w, p + ".conv2")
    if (p + ".conv_shortcut.weight") in w:
        x = conv(x, w, p + ".conv_shortcut", padding=0)
    return (x + h) / 1.0


def attention(x: T, ctx: T, w, p: str, heads: int) -> T:
    """CrossAttention: no bias on q/k/v, bias on to_out.0, softmax(q k^T d^-1/2) v."""
    q = linear(x, w, p + ".to_q")
    k = linear(ctx, w, p + ".to_k")
    v = linear(ctx, w, p + ".to_v")
    B, S, C = q.shape
    d = C // heads

    def split(t):
        return t.view(B, -1, heads, d).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    att = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, S, C)
    return linear(o, w, p + ".to_out.0")


def transformer(x: T, ctx: T, w, p: str, heads: int, groups: int, linear_proj: bool, gelu_approximate: str = "none") -> T:
    """``gelu_approximate='tanh'``: the graph after scripts/optimize.py --approximate-gelu (com.microsoft.FastGelu)."""
    B, C, H, W = x.shape
    r = x
    h = group_norm(x, w, p + ".norm", groups, 1e-6)
    if not linear_proj:
        h = conv(h, w, p + ".proj_in", padding=0)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    else:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = linear(h, w, p + ".proj_in")
    b = p + ".transformer_blocks.0"
    ln = lambda t, n: F.layer_norm(t, (C,), w[f"{b}.{n}.weight"], w[f"{b}.{n}.bias"], 1e-5)
    n1 = ln(h, "norm1")
    h = h + attention(n1, n1, w, b + ".attn1", heads)
    h = h + attention(ln(h, "norm2"), ctx, w, b + ".attn2", heads)
    y = linear(ln(h, "norm3"), w, b + ".ff.net.0.proj")
    a, g = y.chunk(2, dim=-1)
    h = h + linear(a * F.gelu(g, approximate=gelu_approximate), w, b + ".ff.net.2")
    if not linear_proj:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = conv(h, w, p + ".proj_out", padding=0)
    else:
        h = linear(h, w, p + ".proj_out")
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + r


# --------------------------------------------------------------------------- UNet2DConditionModel


def unet_forward(w: Dict[str, T], cfg: dict, sample: T, timestep, ehs: T, taps: dict | None = None) -> T:
    """sample f32 [N,4,h,w], timestep scalar/[1]/[N] (cast to int64 = truncation, sd2pyke.py:121),
    ehs f32 [N,S,D] → f32 [N,4,h,w]."""
    N = sample.shape[0]
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    boc: Sequence[int] = cfg["block_out_channels"]
    lpb = cfg["layers_per_block"]
    lin = cfg["use_linear_projection"]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).to(torch.int64)
    t = t.expand(N) if t.numel() == 1 else t
    temb = timestep_embedding(t, boc[0])
    emb = linear(F.silu(linear(temb, w, "time_embedding.linear_1")), w, "time_embedding.linear_2")

    h = conv(sample.float(), w, "conv_in")
    skips: List[T] = [h]
    for i in range(len(boc)):
        for j in range(lpb):
            h = resnet(h, emb, w, f"down_blocks.{i}.resnets.{j}", groups, eps)
            if cfg["down_attn"][i]:
                h = transformer(h, ehs, w, f"down_blocks.{i}.attentions.{j}", heads_for(cfg, i), groups, lin, cfg.get("gelu_approximate", "none"))
            skips.append(h)
        if i < len(boc) - 1:
            h = conv(h, w, f"down_blocks.{i}.downsamplers.0.conv", stride=2, padding=1)
            skips.append(h)
    if taps is not None:
        taps["down"] = h.clone()
    h = resnet(h, emb, w, "mid_block.resnets.0", groups, eps)
    h = transformer(h, ehs, w, "mid_block.attentions.0", heads_for(cfg, len(boc) - 1), groups, lin, cfg.get("gelu_approximate", "none"))
    h = resnet(h, emb, w, "mid_block.resnets.1", groups, eps)
    if taps is not None:
        taps["mid"] = h.clone()
    up_attn = tuple(reversed(cfg["down_attn"]))
    for i in range(len(boc)):
        for j in range(lpb + 1):
            s = skips.pop()
            h = resnet(torch.cat([h, s], dim=1), emb, w, f"up_blocks.{i}.resnets.{j}", groups, eps)
            if up_attn[i]:
                h = transformer(h, ehs, w, f"up_blocks.{i}.attentions.{j}", heads_for(cfg, len(boc) - 1 - i), groups, lin, cfg.get("gelu_approximate", "none"))
        if i < len(boc) - 1:
            # nearest 2x unless the skip connection dictates another size (forward_upsample_size)
            tgt = skips[-1].shape[-2:]
            if tgt[0] == h.shape[-2] * 2 and tgt[1] == h.shape[-1] * 2:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            else:
                h = F.interpolate(h, size=tuple(tgt), mode="nearest")
            h = conv(h, w, f"up_blocks.{i}.upsamplers.0.conv")
    h = F.silu(group_norm(h, w, "conv_norm_out", groups, eps))
    return conv(h, w, "conv_out")


# --------------------------------------------------------------------------- AutoencoderKL.decode


def default_vae_config(**over) -> dict:
    cfg = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
               norm_num_groups=32)
    cfg.update(over)
    return cfg


def vae_attention_block(x: T, w, p: str, groups: int) -> T:
    B, C, H, W = x.shape
    r = x
    h = group_norm(x, w, p + ".group_norm", groups, 1e-6)
    h = h.view(B, C, H * W).transpose(1, 2)
    q, k, v = linear(h, w, p + ".query"), linear(h, w, p + ".key"), linear(h, w, p + ".value")
    att = torch.softmax((q @ k.transpose(-1, -2)) * (C ** -0.5), dim=-1)
    h = linear(att @ v, w, p + ".proj_attn")
    h = h.transpose(1, 2).reshape(B, C, H, W)
    return (h + r) / 1.0


def vae_decode(w: Dict[str, T], cfg: dict, z: T, taps: dict | None = None) -> T:
    """latent_sample f32 [B,4,h,w] → sample f32 [B,3,8h,8w] (for 4 up blocks)."""
    groups = cfg["norm_num_groups"]
    boc = list(reversed(cfg["block_out_channels"]))  # decoder runs widest first
    z = conv(z.float(), w, "post_quant_conv", padding=0)
    h = conv(z, w, "decoder.conv_in")
    h = resnet(h, None, w, "decoder.mid_block.resnets.0", groups, 1e-6)
    h = vae_attention_block(h, w, "decoder.mid_block.attentions.0", groups)
    h = resnet(h, None, w, "decoder.mid_block.resnets.1", groups, 1e-6)
    if taps is not None:
        taps["mid"] = h.clone()
    for i in range(len(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet(h, None, w, f"decoder.up_blocks.{i}.resnets.{j}", groups, 1e-6)
        if i < len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv(h, w, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    h = F.silu(group_norm(h, w, "decoder.conv_norm_out", groups, 1e-6))
    return conv(h, w, "decoder.conv_out")


def decode_latents(w: Dict[str, T], cfg: dict, latents: T) -> T:
    """impl_main.rs:398-413 + to_image :377-382 → f32 [B,H,W,3] in [0,1].
    ``1/0.18215`` is hard-coded in the reference (the TOML scale-factor is ignored)."""
    z = (torch.tensor(1.0, dtype=torch.float32) / torch.tensor(0.18215, dtype=torch.float32)) * latents.float()
    imgs = []
    for b in range(z.shape[0]):  # batch-1 calls, as the reference does
        img = vae_decode(w, cfg, z[b:b + 1])
        img = img.permute(0, 2, 3, 1) / 2.0 + 0.5
        imgs.append(img.clamp(0.0, 1.0))
    return torch.cat(imgs, dim=0)
