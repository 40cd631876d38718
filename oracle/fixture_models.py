"""ORACLE / TEST INFRASTRUCTURE — not part of the shipped product path.

Independent torch-CPU restatements of the two OTHER graphs the reference ships as test fixtures
(/root/reference/tests/stable-diffusion/{text_encoder,vae_encoder}.onnx; real torch-1.13 / opset-15 exports
of a tiny random Stable Diffusion produced by /root/reference/scripts/sd2pyke.py:202-247,325-346).  They
exist to pin oracle/onnx_interp.py's OPERATOR SEMANTICS on graphs this repo did not write: the interpreter
must reproduce, op by op, what these functional restatements compute from the same initializers
(decomposed LayerNorm, quick-GELU, causal-masked multi-head attention, asymmetric Pad + stride-2 Conv,
Slice / Clip / Exp / RandomNormalLike of the VAE posterior).  Only then is the interpreter used as the second
route from weights to UNet output (tests/test_oracle_unet_graph.py).

The layer arithmetic is third-party (transformers' CLIPTextModel, diffusers==0.14.0 AutoencoderKL.encode,
/root/reference/requirements.txt:3,5), restated from the published architectures.  vae_encoder() is also the
img2img building block the reference only stubs (src/pipelines/stable_diffusion/impl_img2img.rs:177-217).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import model_dir, onnx_io, unet_ref

T = torch.Tensor


def bind_all(model: onnx_io.Model) -> Dict[str, T]:
    """model_dir.bind_weights + the parameters torch's exporter de-duplicated: a `.weight` / `.bias` name that is the OUTPUT of
    an Identity node takes the initializer behind it."""
    w = model_dir.bind_weights(model)
    alias = {n.outputs[0]: n.inputs[0] for n in model.nodes if n.op_type == "Identity"}
    for out in alias:
        if out.endswith(".weight") or out.endswith(".bias"):
            src = out
            for _ in range(64):
                if src in model.initializers or src not in alias:
                    break
                src = alias[src]
            if src in model.initializers:
                w[out] = model.weight(src).astype(np.float32)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}


# --------------------------------------------------------------------------- CLIPTextModel (text_encoder.onnx)


def clip_text_encoder(w: Dict[str, T], input_ids: T, heads: int):
    """transformers CLIPTextModel.forward(input_ids) -> (last_hidden_state [B,S,H], pooler_output [B,H]).
    quick_gelu activation, causal mask, pre-LayerNorm blocks, final_layer_norm; pooled = hidden state at argmax(input_ids)."""
    p = "text_model."
    ids = input_ids.long()
    B, S = ids.shape
    x = w[p + "embeddings.token_embedding.weight"][ids] + w[p + "embeddings.position_embedding.weight"][:S][None]
    H = x.shape[-1]
    d = H // heads
    mask = torch.full((S, S), float(torch.finfo(torch.float32).min)).triu(1)
    ln = lambda t, n: F.layer_norm(t, (H,), w[n + ".weight"], w[n + ".bias"], 1e-5)
    lin = lambda t, n: F.linear(t, w[n + ".weight"], w[n + ".bias"])
    i = 0
    while f"{p}encoder.layers.{i}.self_attn.q_proj.weight" in w:
        b = f"{p}encoder.layers.{i}"
        r = x
        h = ln(x, b + ".layer_norm1")
        q = lin(h, b + ".self_attn.q_proj") * (d ** -0.5)
        k, v = lin(h, b + ".self_attn.k_proj"), lin(h, b + ".self_attn.v_proj")
        sp = lambda t: t.view(B, S, heads, d).transpose(1, 2)
        a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) + mask, dim=-1)
        o = (a @ sp(v)).transpose(1, 2).reshape(B, S, H)
        x = r + lin(o, b + ".self_attn.out_proj")
        r = x
        h = lin(ln(x, b + ".layer_norm2"), b + ".mlp.fc1")
        h = h * torch.sigmoid(1.702 * h)
        x = r + lin(h, b + ".mlp.fc2")
        i += 1
    x = ln(x, p + "final_layer_norm")
    pooled = x[torch.arange(B), ids.argmax(dim=-1)]
    return x, pooled


# --------------------------------------------------------------------------- AutoencoderKL.encode (vae_encoder.onnx)


def vae_encoder(w: Dict[str, T], sample: T, noise: T | None, groups: int = 32) -> T:
    """AutoencoderKLIOWrapper.encode -> latent_dist.sample() (sd2pyke.py:181-187): conv_in, down blocks (resnets, then
    pad (0,1,0,1) + 3x3 stride-2 conv), mid block (resnet, single-head attention, resnet), GroupNorm + SiLU + conv_out, quant_conv
    (1x1) -> (mean, logvar); logvar clamped to [-30, 20]; sample = mean + exp(0.5 logvar) * noise.  GroupNorm eps 1e-6."""
    h = unet_ref.conv(sample.float(), w, "encoder.conv_in")
    i = 0
    while f"encoder.down_blocks.{i}.resnets.0.conv1.weight" in w:
        j = 0
        while f"encoder.down_blocks.{i}.resnets.{j}.conv1.weight" in w:
            h = unet_ref.resnet(h, None, w, f"encoder.down_blocks.{i}.resnets.{j}", groups, 1e-6)
            j += 1
        if f"encoder.down_blocks.{i}.downsamplers.0.conv.weight" in w:
            h = F.pad(h, (0, 1, 0, 1))
            h = unet_ref.conv(h, w, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
        i += 1
    h = unet_ref.resnet(h, None, w, "encoder.mid_block.resnets.0", groups, 1e-6)
    h = unet_ref.vae_attention_block(h, w, "encoder.mid_block.attentions.0", groups)
    h = unet_ref.resnet(h, None, w, "encoder.mid_block.resnets.1", groups, 1e-6)
    h = F.silu(unet_ref.group_norm(h, w, "encoder.conv_norm_out", groups, 1e-6))
    h = unet_ref.conv(h, w, "encoder.conv_out")
    moments = unet_ref.conv(h, w, "quant_conv", padding=0)
    mean, logvar = moments.chunk(2, dim=1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    return mean if noise is None else mean + std * noise
