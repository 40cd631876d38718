"""ORACLE / TEST INFRASTRUCTURE — not part of the shipped product path.

Reads a pyke-diffusers model directory the way ``StableDiffusionPipeline::new`` does
(/root/reference/src/pipelines/stable_diffusion/impl_main.rs:109-181, schema
/root/reference/src/config.rs:17-117) and binds the ONNX initializers of ``unet.onnx`` /
``vae_decoder.onnx`` back to diffusers parameter names for oracle/unet_ref.py.

Binding rules (raw torch-1.13 / opset-15 export dialect, SURVEY.md §3.4, verified on the
reference fixture tests/stable-diffusion/vae_decoder.onnx):
  * initializers named ``*.weight`` / ``*.bias`` bind directly;
  * ``MatMul`` node ``/<module path>/MatMul`` with an initializer 2nd input ⇒ ``<module>.weight`` = init.T;
  * ``Mul`` / ``Add`` node ``/<module path>/{Mul,Add}`` with a [C,1,1] initializer ⇒ GroupNorm gamma / beta;
  * ``Identity`` chains (exporter de-duplication) are resolved to their source initializer.
The product's C++ loader (diffusers_amd/csrc/loader.cpp) implements the same rules independently.
"""
from __future__ import annotations

import os
import re
from typing import Dict, Tuple

import numpy as np

from . import onnx_io

try:
    import tomllib as _toml  # py311+
except ModuleNotFoundError:  # pragma: no cover
    import tomli as _toml


def read_config(root: str) -> dict:
    with open(os.path.join(root, "pyke-diffusers.toml"), "rb") as f:
        cfg = _toml.load(f)
    if cfg.get("pipeline") != "stable-diffusion":
        raise ValueError("not a stable diffusion pipeline")  # impl_main.rs:121
    if cfg.get("framework", {}).get("type") != "orte":
        raise ValueError("bad framework")  # impl_main.rs:113-118 (panics there)
    return cfg


def _module_of(node_name: str) -> str:
    # "/decoder/mid_block/attentions.0/query/MatMul" -> "decoder.mid_block.attentions.0.query"
    parts = [p for p in node_name.split("/") if p]
    return ".".join(parts[:-1])


def bind_weights(model: onnx_io.Model) -> Dict[str, np.ndarray]:
    alias: Dict[str, str] = {}
    for n in model.nodes:
        if n.op_type == "Identity":
            alias[n.outputs[0]] = n.inputs[0]

    def resolve(name: str):
        seen = 0
        while name in alias and seen < 64:
            name = alias[name]
            seen += 1
        return name if name in model.initializers else None

    out: Dict[str, np.ndarray] = {}
    for name in model.initializers:
        if name.endswith(".weight") or name.endswith(".bias"):
            out[name] = model.weight(name).astype(np.float32)
    for n in model.nodes:
        if n.op_type == "MatMul" and len(n.inputs) == 2:
            src = resolve(n.inputs[1])
            if src is not None and len(model.initializers[src].dims) == 2 and n.name.endswith("/MatMul"):
                out[_module_of(n.name) + ".weight"] = np.ascontiguousarray(model.weight(src).astype(np.float32).T)
        elif n.op_type == "Gemm":
            src = resolve(n.inputs[1])
            if src is not None and not n.attr_i("transB", 0):
                out[_module_of(n.name) + ".weight"] = np.ascontiguousarray(model.weight(src).astype(np.float32).T)
        elif n.op_type in ("Mul", "Add") and re.search(r"/(Mul|Add)$", n.name):
            for inp in n.inputs:
                src = resolve(inp)
                if src is None:
                    continue
                dims = model.initializers[src].dims
                if len(dims) == 3 and dims[1] == 1 and dims[2] == 1:
                    key = _module_of(n.name) + (".weight" if n.op_type == "Mul" else ".bias")
                    out[key] = model.weight(src).astype(np.float32).reshape(-1)
    return out


def _attn_head_dim(model: onnx_io.Model, scope_prefix: str) -> int | None:
    """softmax scale = d^-1/2 is a scalar Constant feeding a Mul inside the attention scope."""
    consts = {}
    for n in model.nodes:
        if n.op_type == "Constant" and "value" in n.attrs and n.attrs["value"].t is not None:
            consts[n.outputs[0]] = n.attrs["value"].t
    for n in model.nodes:
        if n.op_type == "Mul" and n.name.startswith(scope_prefix):
            for inp in n.inputs:
                t = consts.get(inp) or model.initializers.get(inp)
                if t is not None and int(np.prod(t.dims or [1])) == 1 and t.data_type in (1, 10):
                    s = float(t.numpy(model.base_dir).reshape(-1)[0])
                    if 0 < s < 1:
                        return int(round(1.0 / (s * s)))
    return None


def infer_unet_config(w: Dict[str, np.ndarray], model: onnx_io.Model | None = None) -> dict:
    nb = 0
    while f"down_blocks.{nb}.resnets.0.conv1.weight" in w:
        nb += 1
    boc = tuple(int(w[f"down_blocks.{i}.resnets.0.conv1.weight"].shape[0]) for i in range(nb))
    lpb = 0
    while f"down_blocks.0.resnets.{lpb}.conv1.weight" in w:
        lpb += 1
    down_attn = tuple(f"down_blocks.{i}.attentions.0.norm.weight" in w for i in range(nb))
    k = "mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight"
    ctx = int(w[k].shape[1])
    lin = w["mid_block.attentions.0.proj_in.weight"].ndim == 2
    heads = []
    for i in range(nb):
        d = None
        if model is not None:
            scope = f"/down_blocks.{i}/attentions.0/transformer_blocks.0/attn1/" if down_attn[i] else "/mid_block/attentions.0/transformer_blocks.0/attn1/"
            d = _attn_head_dim(model, scope)
        if d is None:
            d = 64 if ctx == 1024 else boc[i] // 8
        heads.append(boc[i] // d)
    nh = heads[0] if all(h == heads[0] for h in heads) else tuple(heads)
    return dict(in_channels=int(w["conv_in.weight"].shape[1]), out_channels=int(w["conv_out.weight"].shape[0]),
                block_out_channels=boc, layers_per_block=lpb, down_attn=down_attn, cross_attention_dim=ctx,
                num_heads=nh, use_linear_projection=lin, norm_num_groups=32, norm_eps=1e-5)


def infer_vae_config(w: Dict[str, np.ndarray]) -> dict:
    nb = 0
    while f"decoder.up_blocks.{nb}.resnets.0.conv1.weight" in w:
        nb += 1
    boc = tuple(reversed([int(w[f"decoder.up_blocks.{i}.resnets.0.conv1.weight"].shape[0]) for i in range(nb)]))
    nres = 0
    while f"decoder.up_blocks.0.resnets.{nres}.conv1.weight" in w:
        nres += 1
    return dict(latent_channels=int(w["post_quant_conv.weight"].shape[0]), out_channels=int(w["decoder.conv_out.weight"].shape[0]),
                block_out_channels=boc, layers_per_block=nres - 1, norm_num_groups=32)


def load_unet(root: str) -> Tuple[Dict[str, np.ndarray], dict]:
    cfg = read_config(root)
    m = onnx_io.load(os.path.join(root, cfg["unet"]["path"]))
    w = bind_weights(m)
    return w, infer_unet_config(w, m)


def load_vae_decoder(root: str) -> Tuple[Dict[str, np.ndarray], dict]:
    cfg = read_config(root)
    m = onnx_io.load(os.path.join(root, cfg["vae"]["decoder"]))
    w = bind_weights(m)
    return w, infer_vae_config(w)
