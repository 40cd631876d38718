"""Caller-side tooling: write a pyke-diffusers model directory with SYNTHETIC weights.

There are no Stable Diffusion checkpoints in this environment (and no network), so benches and
parity tests run on random-init weights of the right architecture.  This module writes them in
the same on-disk contract the reference's exporter produces
(/root/reference/scripts/sd2pyke.py:249-362,420-444):

  <root>/pyke-diffusers.toml          (schema: /root/reference/src/config.rs:17-117)
  <root>/unet.onnx [+ unet_weights.bin when > 2000 MB, sd2pyke.py:274,306-316]
  <root>/vae_decoder.onnx

ONNX dialect = torch-1.13 / opset-15 raw export as observed on the reference's fixture
tests/stable-diffusion/vae_decoder.onnx: Conv weights and all biases keep their
PyTorch names; nn.Linear weights are anonymous ``onnx::MatMul_<n>`` stored [in,out]; GroupNorm
affine parameters are anonymous ``onnx::Mul_<n>`` / ``onnx::Add_<n>`` of shape [C,1,1] behind a
Reshape→InstanceNormalization→Reshape→Mul→Add chain; node names carry the module path.
The raw dialect is a complete, EXECUTABLE opset-15 graph: sinusoidal time projection (Cast to int64 = truncation, sd2pyke.py:121),
head split / merge through Reshape + Transpose, GEGLU as two Slices + the decomposed erf GELU, NCHW <-> token layout changes with
Shape / Gather / Unsqueeze / Concat, and the upsample that Resizes to the SIZE of the next skip connection (the exporter traces on a
64 x 65 latent, sd2pyke.py:287-288).  oracle/onnx_interp.py runs it as the second route from weights to UNet output
(tests/test_oracle_unet_graph.py).  The optimizer-fused dialects (fused=True) remain weight-binding files: their com.microsoft
nodes have no interpreter here.
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------- architecture configs


def unet_config(**over) -> dict:
    cfg = dict(
        in_channels=4, out_channels=4,
        block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
        down_attn=(True, True, True, False),
        cross_attention_dim=768, num_heads=8, use_linear_projection=False,
        norm_num_groups=32, norm_eps=1e-5,
    )
    cfg.update(over)
    return cfg


def sd21_unet_config(**over) -> dict:
    return unet_config(cross_attention_dim=1024, num_heads=(5, 10, 20, 20), use_linear_projection=True, **over)


def vae_config(**over) -> dict:
    cfg = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
               norm_num_groups=32)
    cfg.update(over)
    return cfg


def _heads(cfg, level):
    h = cfg["num_heads"]
    return h if isinstance(h, int) else h[level]


# --------------------------------------------------------------------------- parameter inventories (PyTorch names)


def _resnet_shapes(p, cin, cout, temb_dim, out):
    out[p + ".norm1.weight"] = (cin,)
    out[p + ".norm1.bias"] = (cin,)
    out[p + ".conv1.weight"] = (cout, cin, 3, 3)
    out[p + ".conv1.bias"] = (cout,)
    if temb_dim:
        out[p + ".time_emb_proj.weight"] = (cout, temb_dim)
        out[p + ".time_emb_proj.bias"] = (cout,)
    out[p + ".norm2.weight"] = (cout,)
    out[p + ".norm2.bias"] = (cout,)
    out[p + ".conv2.weight"] = (cout, cout, 3, 3)
    out[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        out[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[p + ".conv_shortcut.bias"] = (cout,)


def _transformer_shapes(p, c, ctx, linear_proj, out):
    out[p + ".norm.weight"] = (c,)
    out[p + ".norm.bias"] = (c,)
    out[p + ".proj_in.weight"] = (c, c) if linear_proj else (c, c, 1, 1)
    out[p + ".proj_in.bias"] = (c,)
    b = p + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        out[f"{b}.{n}.weight"] = (c,)
        out[f"{b}.{n}.bias"] = (c,)
    for a, kdim in (("attn1", c), ("attn2", ctx)):
        out[f"{b}.{a}.to_q.weight"] = (c, c)
        out[f"{b}.{a}.to_k.weight"] = (c, kdim)
        out[f"{b}.{a}.to_v.weight"] = (c, kdim)
        out[f"{b}.{a}.to_out.0.weight"] = (c, c)
        out[f"{b}.{a}.to_out.0.bias"] = (c,)
    out[f"{b}.ff.net.0.proj.weight"] = (8 * c, c)
    out[f"{b}.ff.net.0.proj.bias"] = (8 * c,)
    out[f"{b}.ff.net.2.weight"] = (c, 4 * c)
    out[f"{b}.ff.net.2.bias"] = (c,)
    out[p + ".proj_out.weight"] = (c, c) if linear_proj else (c, c, 1, 1)
    out[p + ".proj_out.bias"] = (c,)


def unet_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """diffusers-0.14 UNet2DConditionModel parameter names → shapes."""
    out: "OrderedDict[str, tuple]" = OrderedDict()
    boc = list(cfg["block_out_channels"])
    lpb, ctx, lin = cfg["layers_per_block"], cfg["cross_attention_dim"], cfg["use_linear_projection"]
    temb = boc[0] * 4
    out["conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3)
    out["conv_in.bias"] = (boc[0],)
    out["time_embedding.linear_1.weight"] = (temb, boc[0])
    out["time_embedding.linear_1.bias"] = (temb,)
    out["time_embedding.linear_2.weight"] = (temb, temb)
    out["time_embedding.linear_2.bias"] = (temb,)
    skips = [boc[0]]
    c = boc[0]
    for i, co in enumerate(boc):
        for j in range(lpb):
            _resnet_shapes(f"down_blocks.{i}.resnets.{j}", c, co, temb, out)
            c = co
            if cfg["down_attn"][i]:
                _transformer_shapes(f"down_blocks.{i}.attentions.{j}", c, ctx, lin, out)
            skips.append(c)
        if i < len(boc) - 1:
            out[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            out[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
            skips.append(c)
    _resnet_shapes("mid_block.resnets.0", c, c, temb, out)
    _transformer_shapes("mid_block.attentions.0", c, ctx, lin, out)
    _resnet_shapes("mid_block.resnets.1", c, c, temb, out)
    up_attn = tuple(reversed(cfg["down_attn"]))
    for i, co in enumerate(reversed(boc)):
        for j in range(lpb + 1):
            s = skips.pop()
            _resnet_shapes(f"up_blocks.{i}.resnets.{j}", c + s, co, temb, out)
            c = co
            if up_attn[i]:
                _transformer_shapes(f"up_blocks.{i}.attentions.{j}", c, ctx, lin, out)
        if i < len(boc) - 1:
            out[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            out[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    out["conv_norm_out.weight"] = (c,)
    out["conv_norm_out.bias"] = (c,)
    out["conv_out.weight"] = (cfg["out_channels"], c, 3, 3)
    out["conv_out.bias"] = (cfg["out_channels"],)
    return out


def vae_decoder_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    out: "OrderedDict[str, tuple]" = OrderedDict()
    boc = list(reversed(cfg["block_out_channels"]))
    lc = cfg["latent_channels"]
    out["post_quant_conv.weight"] = (lc, lc, 1, 1)
    out["post_quant_conv.bias"] = (lc,)
    out["decoder.conv_in.weight"] = (boc[0], lc, 3, 3)
    out["decoder.conv_in.bias"] = (boc[0],)
    c = boc[0]
    _resnet_shapes("decoder.mid_block.resnets.0", c, c, 0, out)
    a = "decoder.mid_block.attentions.0"
    out[a + ".group_norm.weight"] = (c,)
    out[a + ".group_norm.bias"] = (c,)
    for n in ("query", "key", "value", "proj_attn"):
        out[f"{a}.{n}.weight"] = (c, c)
        out[f"{a}.{n}.bias"] = (c,)
    _resnet_shapes("decoder.mid_block.resnets.1", c, c, 0, out)
    for i, co in enumerate(boc):
        for j in range(cfg["layers_per_block"] + 1):
            _resnet_shapes(f"decoder.up_blocks.{i}.resnets.{j}", c, co, 0, out)
            c = co
        if i < len(boc) - 1:
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    out["decoder.conv_norm_out.weight"] = (c,)
    out["decoder.conv_norm_out.bias"] = (c,)
    out["decoder.conv_out.weight"] = (cfg["out_channels"], c, 3, 3)
    out["decoder.conv_out.bias"] = (cfg["out_channels"],)
    return out


def vae_encoder_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    """AutoencoderKL encoder + quant_conv (diffusers 0.14.0 names, as the reference's vae_encoder.onnx fixture carries them)."""
    out: "OrderedDict[str, tuple]" = OrderedDict()
    boc = list(cfg["block_out_channels"])
    lc = cfg["latent_channels"]
    out["encoder.conv_in.weight"] = (boc[0], cfg["out_channels"], 3, 3)
    out["encoder.conv_in.bias"] = (boc[0],)
    c = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            _resnet_shapes(f"encoder.down_blocks.{i}.resnets.{j}", c, co, 0, out)
            c = co
        if i < len(boc) - 1:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    _resnet_shapes("encoder.mid_block.resnets.0", c, c, 0, out)
    a = "encoder.mid_block.attentions.0"
    out[a + ".group_norm.weight"] = (c,)
    out[a + ".group_norm.bias"] = (c,)
    for n in ("query", "key", "value", "proj_attn"):
        out[f"{a}.{n}.weight"] = (c, c)
        out[f"{a}.{n}.bias"] = (c,)
    _resnet_shapes("encoder.mid_block.resnets.1", c, c, 0, out)
    out["encoder.conv_norm_out.weight"] = (c,)
    out["encoder.conv_norm_out.bias"] = (c,)
    out["encoder.conv_out.weight"] = (2 * lc, c, 3, 3)
    out["encoder.conv_out.bias"] = (2 * lc,)
    out["quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1)
    out["quant_conv.bias"] = (2 * lc,)
    return out


def random_weights(shapes: Dict[str, tuple], seed: int, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Random-init weights (fan-in scaled so activations stay O(1) through the residual stacks):
    conv/linear ~ N(0, gain/sqrt(fan_in)), norm gamma ~ 1 + 0.1 N, biases/betas ~ 0.05 N."""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith(".bias"):
            w = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.05)
        elif len(shp) == 1:
            w = np.float32(1.0) + rng.standard_normal(shp, dtype=np.float32) * np.float32(0.1)
        else:
            fan_in = int(np.prod(shp[1:]))
            w = rng.standard_normal(shp, dtype=np.float32)
            w *= np.float32(gain / np.sqrt(fan_in))
        out[name] = w
    return out


# --------------------------------------------------------------------------- protobuf writer


def _vi(x: int) -> bytes:
    if x < 0:
        x += 1 << 64
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fn: int, payload: bytes) -> bytes:
    return _vi((fn << 3) | 2) + _vi(len(payload)) + payload


def _iv(fn: int, v: int) -> bytes:
    return _vi(fn << 3) + _vi(v)


_NP2ONNX = {np.dtype(np.float32): 1, np.dtype(np.int64): 7, np.dtype(np.float16): 10}


def _tensor(name: str, arr: np.ndarray, external: Tuple[str, int] | None = None) -> bytes:
    arr = np.asarray(arr).copy(order="C")          # (np.ascontiguousarray would turn a 0-d scalar into shape [1])
    b = b"".join(_iv(1, int(d)) for d in arr.shape)
    b += _iv(2, _NP2ONNX[arr.dtype]) + _ld(8, name.encode())
    if external is None:
        b += _ld(9, arr.tobytes())
    else:
        loc, off = external
        for k, v in (("location", loc), ("offset", str(off)), ("length", str(arr.nbytes))):
            b += _ld(13, _ld(1, k.encode()) + _ld(2, v.encode()))
        b += _iv(14, 1)
    return b


def _attr_i(name, v):
    return _ld(5, _ld(1, name.encode()) + _iv(3, v) + _iv(20, 2))


def _attr_f(name, v):
    return _ld(5, _ld(1, name.encode()) + _vi((2 << 3) | 5) + struct.pack("<f", v) + _iv(20, 1))


def _attr_ints(name, vs):
    return _ld(5, _ld(1, name.encode()) + b"".join(_iv(8, int(v)) for v in vs) + _iv(20, 7))


def _attr_s(name, s):
    return _ld(5, _ld(1, name.encode()) + _ld(4, s.encode()) + _iv(20, 3))


def _attr_t(name, tensor_bytes):
    return _ld(5, _ld(1, name.encode()) + _ld(5, tensor_bytes) + _iv(20, 4))


def _node(op, name, inputs, outputs, attrs=b"", domain=""):
    b = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    return _ld(3, name.encode()) + _ld(4, op.encode()) + b + attrs + (_ld(7, domain.encode()) if domain else b"")


class GraphWriter:
    """Accumulates nodes/initializers; weights above ``external_threshold`` bytes total go to one
    external file, as sd2pyke.py:306-316 does for > 2000 MB UNets."""

    def __init__(self, dtype=np.float32, fused: bool = False, approximate_gelu: bool = False, nv: bool = False):
        """``fused``: emit the graph the reference's optimizer leaves behind (scripts/optimize.py --preset gpu:
        com.microsoft Attention / MultiHeadAttention with packed KV / GroupNorm, LayerNormalization, Gelu or FastGelu) instead
        of the raw torch export.  Weight packing follows scripts/_optimizer/fusion_attention_unet.py:215-226,231-280,
        fusion_attention_vae.py:160-190, fusion_group_norm.py:108-160."""
        self.nodes: List[bytes] = []
        self.inits: List[Tuple[str, np.ndarray]] = []
        self.counter = 100
        self.dtype = np.dtype(dtype)
        self.val = 0
        self.cidx = 0
        self.fused = fused
        self.approximate_gelu = approximate_gelu
        # ``nv``: additionally the fusions of the `*-nv` presets (scripts/optimize.py:102-130): NhwcConv behind weight / activation
        # Transposes (fusion_nhwc_conv.py:35-90, update_weight=False), BiasSplitGelu (fusion_biassplitgelu.py:88-112), BiasAdd
        # (fusion_bias_add.py:30-60) and SkipLayerNormalization for the residual add in front of norm2 / norm3
        self.nv = nv and fused
        self.fcount: Dict[str, int] = {}

    def fname(self, prefix: str) -> str:
        """OnnxModel.create_node_name: <prefix>_<running index>."""
        i = self.fcount.get(prefix, 0)
        self.fcount[prefix] = i + 1
        return f"{prefix}_{i}"

    def anon(self, kind: str) -> str:
        self.counter += 1
        return f"onnx::{kind}_{self.counter}"

    def tmp(self, scope: str, op: str) -> str:
        self.val += 1
        return f"{scope}/{op}_output_{self.val}"

    def init(self, name: str, arr: np.ndarray):
        if arr.dtype == np.float32 and self.dtype != np.float32:
            arr = arr.astype(self.dtype)
        self.inits.append((name, arr))

    def node(self, op, name, inputs, outputs, attrs=b"", domain=""):
        self.nodes.append(_node(op, name, inputs, outputs, attrs, domain))

    def const(self, scope, value: np.ndarray) -> str:
        self.cidx += 1
        out = f"{scope}/Constant_{self.cidx}_output_0"
        self.node("Constant", f"{scope}/Constant_{self.cidx}", [], [out], _attr_t("value", _tensor("", value)))
        return out

    # --- module emitters (dialect of the fixture) ---
    def conv(self, scope, pname, w, x, stride=1, k=3, pad=None):
        out = self.tmp(scope, "Conv")
        self.init(pname + ".weight", w[pname + ".weight"])
        self.init(pname + ".bias", w[pname + ".bias"])
        if pad is None:
            pad = 1 if k == 3 else 0
        attrs = (_attr_ints("dilations", [1, 1]) + _attr_i("group", 1) + _attr_ints("kernel_shape", [k, k]) +
                 _attr_ints("pads", [pad] * 4) + _attr_ints("strides", [stride, stride]))
        if self.nv:
            xin = x + "_NHWC_in"
            self.node("Transpose", self.fname("Transpose_NCHW_to_NHWC"), [x], [xin], _attr_ints("perm", [0, 2, 3, 1]))
            wt = pname + ".weight_NHWC"
            self.node("Transpose", self.fname("Transpose_NCHW_to_NHWC"), [pname + ".weight"], [wt], _attr_ints("perm", [0, 2, 3, 1]))
            name = self.fname("NhwcConv")
            self.node("NhwcConv", f"{name}-{scope}/Conv", [xin, wt, pname + ".bias"], [f"{name}_out-{out}"], attrs, "com.microsoft")
            self.node("Transpose", self.fname("Transpose_NHWC_to_NCHW"), [f"{name}_out-{out}"], [out], _attr_ints("perm", [0, 3, 1, 2]))
            return out
        self.node("Conv", f"{scope}/Conv", [x, pname + ".weight", pname + ".bias"], [out], attrs)
        return out

    def linear(self, scope, pname, w, x, bias=True):
        """3-D input ⇒ MatMul with anonymous [in,out] weight + Add(bias, matmul) (fixture dialect)."""
        wn = self.anon("MatMul")
        self.init(wn, np.ascontiguousarray(w[pname + ".weight"].T))
        out = self.tmp(scope, "MatMul")
        self.node("MatMul", f"{scope}/MatMul", [x, wn], [out])
        if bias:
            self.init(pname + ".bias", w[pname + ".bias"])
            out2 = self.tmp(scope, "Add")
            self.node("Add", f"{scope}/Add", [pname + ".bias", out], [out2])
            out = out2
        return out

    def gemm(self, scope, pname, w, x):
        """2-D input ⇒ Gemm(transB=1) with the PyTorch-named [out,in] weight (time-embedding MLP)."""
        self.init(pname + ".weight", w[pname + ".weight"])
        self.init(pname + ".bias", w[pname + ".bias"])
        out = self.tmp(scope, "Gemm")
        self.node("Gemm", f"{scope}/Gemm", [x, pname + ".weight", pname + ".bias"], [out],
                  _attr_f("alpha", 1.0) + _attr_f("beta", 1.0) + _attr_i("transB", 1))
        return out

    def group_norm(self, scope, pname, w, x, eps, groups=32, swish=False):
        if self.fused:
            # fusion_group_norm.py: Transpose(NCHW->NHWC) -> com.microsoft.GroupNorm(gamma, beta; activation = swish) -> Transpose
            name = self.fname("GroupNorm")
            self.init(name + "_gamma", w[pname + ".weight"].reshape(-1))
            self.init(name + "_beta", w[pname + ".bias"].reshape(-1))
            self.node("Transpose", self.fname("Transpose_NCHW_to_NHWC"), [x], [x + "_NHWC"], _attr_ints("perm", [0, 2, 3, 1]))
            out = self.tmp(scope, "Mul" if swish else "Add")
            self.node("GroupNorm", name, [x + "_NHWC", name + "_gamma", name + "_beta"], [out + "_NHWC"],
                      _attr_f("epsilon", eps) + _attr_i("groups", groups) + _attr_i("activation", 1 if swish else 0), "com.microsoft")
            self.node("Transpose", self.fname("Transpose_NHWC_to_NCHW"), [out + "_NHWC"], [out], _attr_ints("perm", [0, 3, 1, 2]))
            return out
        shp = self.const(scope, np.asarray([0, groups, -1], np.int64))
        r = self.tmp(scope, "Reshape")
        self.node("Reshape", f"{scope}/Reshape", [x, shp], [r])
        one = self.const(scope, np.ones(groups, np.float32))
        zero = self.const(scope, np.zeros(groups, np.float32))
        inn = self.tmp(scope, "InstanceNormalization")
        self.node("InstanceNormalization", f"{scope}/InstanceNormalization", [r, one, zero], [inn], _attr_f("epsilon", eps))
        s = self.tmp(scope, "Shape")
        self.node("Shape", f"{scope}/Shape", [x], [s])
        r1 = self.tmp(scope, "Reshape_1")
        self.node("Reshape", f"{scope}/Reshape_1", [inn, s], [r1])
        g, b = self.anon("Mul"), self.anon("Add")
        self.init(g, w[pname + ".weight"].reshape(-1, 1, 1))
        self.init(b, w[pname + ".bias"].reshape(-1, 1, 1))
        m = self.tmp(scope, "Mul")
        self.node("Mul", f"{scope}/Mul", [r1, g], [m])
        a = self.tmp(scope, "Add")
        self.node("Add", f"{scope}/Add", [m, b], [a])
        return a

    def layer_norm(self, scope, pname, w, x, eps=1e-5):
        """opset-15 decomposed LayerNorm with named gamma/beta."""
        self.init(pname + ".weight", w[pname + ".weight"])
        self.init(pname + ".bias", w[pname + ".bias"])
        if self.fused:   # fusion_layernorm.py: one LayerNormalization node that keeps the named gamma / beta
            out = self.tmp(scope, "Add_1")
            self.node("LayerNormalization", self.fname("LayerNorm"), [x, pname + ".weight", pname + ".bias"], [out],
                      _attr_i("axis", -1) + _attr_f("epsilon", eps))
            return out
        mu = self.tmp(scope, "ReduceMean")
        self.node("ReduceMean", f"{scope}/ReduceMean", [x], [mu], _attr_ints("axes", [-1]))
        d = self.tmp(scope, "Sub")
        self.node("Sub", f"{scope}/Sub", [x, mu], [d])
        p = self.tmp(scope, "Pow")
        self.node("Pow", f"{scope}/Pow", [d, self.const(scope, np.asarray(2.0, np.float32))], [p])
        v = self.tmp(scope, "ReduceMean_1")
        self.node("ReduceMean", f"{scope}/ReduceMean_1", [p], [v], _attr_ints("axes", [-1]))
        ve = self.tmp(scope, "Add")
        self.node("Add", f"{scope}/Add", [v, self.const(scope, np.asarray(eps, np.float32))], [ve])
        sd = self.tmp(scope, "Sqrt")
        self.node("Sqrt", f"{scope}/Sqrt", [ve], [sd])
        q = self.tmp(scope, "Div")
        self.node("Div", f"{scope}/Div", [d, sd], [q])
        m = self.tmp(scope, "Mul")
        self.node("Mul", f"{scope}/Mul", [q, pname + ".weight"], [m])
        a = self.tmp(scope, "Add_1")
        self.node("Add", f"{scope}/Add_1", [m, pname + ".bias"], [a])
        return a

    def skip_layer_norm(self, scope, pname, w, x, skip, eps=1e-5):
        """com.microsoft.SkipLayerNormalization(input, skip, gamma, beta) -> (normalized, _, _, input + skip): what the optimizer's
        skip_layer_norm option leaves for `Add` + LayerNormalization (onnx_model_unet.py:120-121)."""
        self.init(pname + ".weight", w[pname + ".weight"])
        self.init(pname + ".bias", w[pname + ".bias"])
        out, summed = self.tmp(scope, "Add_1"), self.tmp(scope, "SkipSum")
        self.node("SkipLayerNormalization", self.fname("SkipLayerNorm"), [x, skip, pname + ".weight", pname + ".bias"], [out, "", "", summed],
                  _attr_f("epsilon", eps), "com.microsoft")
        return out, summed

    def unary(self, op, scope, x, name=None):
        out = self.tmp(scope, op)
        self.node(op, f"{scope}/{name or op}", [x], [out])
        return out

    def binary(self, op, scope, a, b, name=None):
        out = self.tmp(scope, op)
        self.node(op, f"{scope}/{name or op}", [a, b], [out])
        return out

    def silu(self, scope, x):
        s = self.unary("Sigmoid", scope, x)
        return self.binary("Mul", scope, x, s)

    def i64(self, scope, values) -> str:
        return self.const(scope, np.asarray(values, np.int64))

    def op(self, op, scope, inputs, attrs=b"", name=None):
        out = self.tmp(scope, name or op)
        self.val += 0
        self.node(op, f"{scope}/{name or op}_{self.val}", list(inputs), [out], attrs)
        return out

    def transpose(self, scope, x, perm):
        return self.op("Transpose", scope, [x], _attr_ints("perm", perm))

    def reshape(self, scope, x, shape_name):
        return self.op("Reshape", scope, [x, shape_name])

    def dim(self, scope, shape_of, index) -> str:
        """Shape(x)[index] as a 1-element int64 tensor (Shape -> Gather(scalar) -> Unsqueeze, as torch exports `x.shape[i]`)."""
        shp = self.op("Shape", scope, [shape_of])
        g = self.op("Gather", scope, [shp, self.const(scope, np.asarray(index, np.int64))], _attr_i("axis", 0))
        return self.op("Unsqueeze", scope, [g, self.i64(scope, [0])])

    def serialize(self, path: str, inputs: Sequence[str], outputs: Sequence[str], external_file: str | None):
        """Streams the ModelProto to disk piece by piece (no quadratic byte-string concatenation)."""
        base = os.path.dirname(os.path.abspath(path))
        pieces: List = [_ld(1, n) for n in self.nodes]
        pieces.append(_ld(2, b"torch_jit"))
        if external_file:
            off = 0
            with open(os.path.join(base, external_file), "wb") as f:
                for name, arr in self.inits:
                    arr = np.ascontiguousarray(arr)
                    pieces.append(_ld(5, _tensor(name, arr, (external_file, off))))
                    f.write(memoryview(arr).cast("B"))
                    off += arr.nbytes
        else:
            for name, arr in self.inits:
                arr = np.ascontiguousarray(arr)
                head = b"".join(_iv(1, int(d)) for d in arr.shape) + _iv(2, _NP2ONNX[arr.dtype]) + _ld(8, name.encode())
                raw_hdr = _vi((9 << 3) | 2) + _vi(arr.nbytes)
                tlen = len(head) + len(raw_hdr) + arr.nbytes
                pieces.append(_vi((5 << 3) | 2) + _vi(tlen) + head + raw_hdr)
                pieces.append(memoryview(arr).cast("B") if arr.nbytes else b"")
        for n in inputs:
            pieces.append(_ld(11, _ld(1, n.encode())))
        for n in outputs:
            pieces.append(_ld(12, _ld(1, n.encode())))
        glen = sum(len(p) for p in pieces)
        with open(path, "wb") as f:
            f.write(_iv(1, 8) + _ld(2, b"pytorch") + _ld(3, b"1.13.1"))
            f.write(_vi((7 << 3) | 2) + _vi(glen))
            for p in pieces:
                f.write(p)
            f.write(_ld(8, _ld(1, b"") + _iv(2, 15)))


# --------------------------------------------------------------------------- graph emitters


def _emit_resnet(gw: GraphWriter, scope, p, w, x, emb, eps, groups):
    if gw.fused:   # the swish after each GroupNorm is folded into the node (activation = 1)
        h = gw.group_norm(scope + "/norm1", p + ".norm1", w, x, eps, groups, swish=True)
        h = gw.conv(scope + "/conv1", p + ".conv1", w, h)
        if emb is not None and (p + ".time_emb_proj.weight") in w:
            e = gw.silu(scope + "/nonlinearity_1", emb)
            e = gw.gemm(scope + "/time_emb_proj", p + ".time_emb_proj", w, e)
            h = gw.binary("Add", scope, h, e)
        h = gw.group_norm(scope + "/norm2", p + ".norm2", w, h, eps, groups, swish=True)
        h = gw.conv(scope + "/conv2", p + ".conv2", w, h)
        if (p + ".conv_shortcut.weight") in w:
            x = gw.conv(scope + "/conv_shortcut", p + ".conv_shortcut", w, x, k=1)
        return gw.binary("Add", scope, x, h, "Add_1")
    h = gw.group_norm(scope + "/norm1", p + ".norm1", w, x, eps, groups)
    h = gw.silu(scope + "/nonlinearity", h)
    h = gw.conv(scope + "/conv1", p + ".conv1", w, h)
    if emb is not None and (p + ".time_emb_proj.weight") in w:
        e = gw.silu(scope + "/nonlinearity_1", emb)
        e = gw.gemm(scope + "/time_emb_proj", p + ".time_emb_proj", w, e)
        e = gw.op("Unsqueeze", scope, [e, gw.i64(scope, [2, 3])])          # temb[:, :, None, None]
        h = gw.binary("Add", scope, h, e)
    h = gw.group_norm(scope + "/norm2", p + ".norm2", w, h, eps, groups)
    h = gw.silu(scope + "/nonlinearity_2", h)
    h = gw.conv(scope + "/conv2", p + ".conv2", w, h)
    if (p + ".conv_shortcut.weight") in w:
        x = gw.conv(scope + "/conv_shortcut", p + ".conv_shortcut", w, x, k=1)
    s = gw.binary("Add", scope, x, h, "Add_1")
    return gw.binary("Div", scope, s, gw.const(scope, np.asarray(1.0, np.float32)))


def _emit_attention_fused(gw: GraphWriter, scope, p, w, x, ctx, heads, c):
    """fusion_attention_unet.py: self-attention -> com.microsoft.Attention with one [C, 3C] weight (np.stack((q,k,v), axis=1)) and a
    zero bias; cross-attention -> to_q MatMul kept, K/V weights interleaved per head into MatMul_KV_n (np.dstack of [c, n, h]),
    Reshape to [0, 0, n, 2, h], com.microsoft.MultiHeadAttention(query, packed kv)."""
    qw, kw, vw = (np.ascontiguousarray(w[p + f".to_{t}.weight"].T) for t in "qkv")   # [in, out] as MatMul stores them
    d = c // heads
    out = gw.tmp(scope, "Reshape_7")
    if x == ctx:
        name = gw.fname("Attention")
        gw.init(name + "_qkv_weight", np.stack((qw, kw, vw), axis=1).reshape(qw.shape[0], 3 * qw.shape[1]))
        gw.init(name + "_qkv_bias", np.zeros(3 * c, np.float32))
        gw.node("Attention", name, [x, name + "_qkv_weight", name + "_qkv_bias"], [out], _attr_i("num_heads", heads), "com.microsoft")
    else:
        q = gw.linear(scope + "/to_q", p + ".to_q", w, x, bias=False)
        mm = gw.fname("MatMul_KV")
        cin = kw.shape[0]
        gw.init(mm + "_weight", np.dstack([kw.reshape(cin, heads, d), vw.reshape(cin, heads, d)]).reshape(cin, heads * 2 * d))
        gw.node("MatMul", mm, [ctx, mm + "_weight"], [mm + "_out"])
        gw.inits.append((mm + "_reshape_shape", np.asarray([0, 0, heads, 2, d], np.int64)))
        kout = gw.tmp(scope + "/to_k", "MatMul")
        gw.node("Reshape", mm + "_reshape", [mm + "_out", mm + "_reshape_shape"], [kout])
        gw.node("MultiHeadAttention", gw.fname("MultiHeadAttention"), [q, kout], [out], _attr_i("num_heads", heads), "com.microsoft")
    return gw.linear(scope + "/to_out.0", p + ".to_out.0", w, out, bias=True)


def _emit_attention(gw: GraphWriter, scope, p, w, x, ctx, heads, c):
    if gw.fused:
        return _emit_attention_fused(gw, scope, p, w, x, ctx, heads, c)
    q = gw.linear(scope + "/to_q", p + ".to_q", w, x, bias=False)
    k = gw.linear(scope + "/to_k", p + ".to_k", w, ctx, bias=False)
    v = gw.linear(scope + "/to_v", p + ".to_v", w, ctx, bias=False)
    d = c // heads
    # head split: [B, S, C] -> [B, S, heads, d] -> [B, heads, S, d]; scores = q k^T * d^-1/2; merge back to [B, S, C]
    shp = gw.const(scope, np.asarray([0, -1, heads, d], np.int64))
    q = gw.transpose(scope, gw.binary("Reshape", scope, q, shp), [0, 2, 1, 3])
    k = gw.transpose(scope, gw.binary("Reshape", scope, k, shp, "Reshape_1"), [0, 2, 3, 1])
    v = gw.transpose(scope, gw.binary("Reshape", scope, v, shp, "Reshape_2"), [0, 2, 1, 3])
    s = gw.binary("MatMul", scope, q, k)
    s = gw.binary("Mul", scope, s, gw.const(scope, np.asarray(d ** -0.5, np.float32)))
    s = gw.op("Softmax", scope, [s], _attr_i("axis", -1))
    o = gw.binary("MatMul", scope, s, v, "MatMul_1")
    o = gw.transpose(scope, o, [0, 2, 1, 3])
    o = gw.binary("Reshape", scope, o, gw.const(scope, np.asarray([0, -1, c], np.int64)), "Reshape_3")
    return gw.linear(scope + "/to_out.0", p + ".to_out.0", w, o, bias=True)


def _emit_transformer(gw: GraphWriter, scope, p, w, x, ctx, heads, c, lin, groups):
    h = gw.group_norm(scope + "/norm", p + ".norm", w, x, 1e-6, groups)
    raw = not gw.fused
    if raw:
        # NCHW -> tokens [N, H*W, C]: conv projection first (SD1.x), Linear projection after the layout change (SD2.x)
        tok = gw.const(scope, np.asarray([0, -1, c], np.int64))
        if lin:
            h = gw.reshape(scope, gw.transpose(scope, h, [0, 2, 3, 1]), tok)
            h = gw.linear(scope + "/proj_in", p + ".proj_in", w, h)
        else:
            h = gw.conv(scope + "/proj_in", p + ".proj_in", w, h, k=1)
            h = gw.reshape(scope, gw.transpose(scope, h, [0, 2, 3, 1]), tok)
    else:
        h = gw.linear(scope + "/proj_in", p + ".proj_in", w, h) if lin else gw.conv(scope + "/proj_in", p + ".proj_in", w, h, k=1)
    bs, bp = scope + "/transformer_blocks.0", p + ".transformer_blocks.0"
    n = gw.layer_norm(bs + "/norm1", bp + ".norm1", w, h)
    if gw.nv:
        n, h = gw.skip_layer_norm(bs + "/norm2", bp + ".norm2", w, _emit_attention(gw, bs + "/attn1", bp + ".attn1", w, n, n, heads, c), h)
        n, h = gw.skip_layer_norm(bs + "/norm3", bp + ".norm3", w, _emit_attention(gw, bs + "/attn2", bp + ".attn2", w, n, ctx, heads, c), h)
        y = gw.linear(bs + "/ff/net.0/proj", bp + ".ff.net.0.proj", w, n, bias=False)
        gw.init(bp + ".ff.net.0.proj.bias", w[bp + ".ff.net.0.proj.bias"])
        g = gw.tmp(bs + "/ff/net.0", "Mul_3")
        gw.node("BiasSplitGelu", gw.fname("BiasSplitGelu"), [y, bp + ".ff.net.0.proj.bias"], [g], b"", "com.microsoft")
        y = gw.linear(bs + "/ff/net.2", bp + ".ff.net.2", w, g, bias=False)
        gw.init(bp + ".ff.net.2.bias", w[bp + ".ff.net.2.bias"])
        out = gw.tmp(bs, "Add_2")
        gw.node("BiasAdd", gw.fname("BiasAdd"), [y, bp + ".ff.net.2.bias", h], [out], b"", "com.microsoft")
        h = out
    else:
        h = gw.binary("Add", bs, h, _emit_attention(gw, bs + "/attn1", bp + ".attn1", w, n, n, heads, c))
        n = gw.layer_norm(bs + "/norm2", bp + ".norm2", w, h)
        h = gw.binary("Add", bs, h, _emit_attention(gw, bs + "/attn2", bp + ".attn2", w, n, ctx, heads, c), "Add_1")
        n = gw.layer_norm(bs + "/norm3", bp + ".norm3", w, h)
        y = gw.linear(bs + "/ff/net.0/proj", bp + ".ff.net.0.proj", w, n)
        if gw.fused:
            # fusion_gelu.py -> Gelu ; with --approximate-gelu fusion_gelu_approximation.py turns it into com.microsoft.FastGelu (tanh form)
            g = gw.tmp(bs + "/ff/net.0", "Mul_3")
            if gw.approximate_gelu:
                gw.node("FastGelu", gw.fname("FastGelu_Gelu_Approximation"), [y], [g], b"", "com.microsoft")
            else:
                gw.node("Gelu", gw.fname("Gelu"), [y], [g], b"", "com.microsoft")
            y = gw.binary("Mul", bs + "/ff/net.0", y, g)
        else:
            # GEGLU (fusion_biassplitgelu.py:20-32): value = first half, gate = second half; exact-erf GELU on the gate as
            # x -> Div(sqrt 2) -> Erf -> Add(1) -> Mul(x) -> Mul(0.5) (fusion_gelu.py:27-45)
            fs = bs + "/ff/net.0"
            val = gw.op("Slice", fs, [y, gw.i64(fs, [0]), gw.i64(fs, [4 * c]), gw.i64(fs, [-1])])
            gate = gw.op("Slice", fs, [y, gw.i64(fs, [4 * c]), gw.i64(fs, [8 * c]), gw.i64(fs, [-1])], name="Slice_1")
            g = gw.binary("Div", fs, gate, gw.const(bs, np.asarray(1.4142135381698608, np.float32)))
            g = gw.unary("Erf", fs, g)
            g = gw.binary("Add", fs, g, gw.const(bs, np.asarray(1.0, np.float32)))
            g = gw.binary("Mul", fs, gate, g, "Mul_1")
            g = gw.binary("Mul", fs, g, gw.const(bs, np.asarray(0.5, np.float32)), "Mul_2")
            y = gw.binary("Mul", fs, val, g, "Mul_3")
        y = gw.linear(bs + "/ff/net.2", bp + ".ff.net.2", w, y)
        h = gw.binary("Add", bs, h, y, "Add_2")
    if raw:
        # tokens -> NCHW with the batch / height / width of the block input
        nhwc = gw.op("Concat", scope, [gw.dim(scope, x, 0), gw.dim(scope, x, 2), gw.dim(scope, x, 3), gw.i64(scope, [c])], _attr_i("axis", 0))
        if lin:
            h = gw.linear(scope + "/proj_out", p + ".proj_out", w, h)
            h = gw.transpose(scope, gw.reshape(scope, h, nhwc), [0, 3, 1, 2])
        else:
            h = gw.transpose(scope, gw.reshape(scope, h, nhwc), [0, 3, 1, 2])
            h = gw.conv(scope + "/proj_out", p + ".proj_out", w, h, k=1)
    else:
        h = gw.linear(scope + "/proj_out", p + ".proj_out", w, h) if lin else gw.conv(scope + "/proj_out", p + ".proj_out", w, h, k=1)
    return gw.binary("Add", scope, h, x)


def write_unet_onnx(path: str, cfg: dict, w: Dict[str, np.ndarray], dtype=np.float32, external: bool | None = None,
                    fused: bool = False, approximate_gelu: bool = False, nv: bool = False):
    gw = GraphWriter(dtype, fused, approximate_gelu, nv)
    boc = list(cfg["block_out_channels"])
    lpb, lin, groups, eps = cfg["layers_per_block"], cfg["use_linear_projection"], cfg["norm_num_groups"], cfg["norm_eps"]
    if fused:
        t = gw.unary("Cast", "", "timestep")
        t = gw.unary("Sin", "/time_proj", t)  # binding-only file: sinusoid plumbing abbreviated
    else:
        # timestep.long() (sd2pyke.py:121: truncation) -> Timesteps(boc[0], flip_sin_to_cos=True, freq_shift=0): [cos | sin] of t * 10000^(-k/half)
        half = boc[0] // 2
        ti = gw.op("Cast", "", ["timestep"], _attr_i("to", 7))
        tf = gw.op("Cast", "/time_proj", [ti], _attr_i("to", 1))
        tf = gw.op("Unsqueeze", "/time_proj", [tf, gw.i64("/time_proj", [1])])
        freqs = np.exp(-np.log(np.float32(10000.0)) * np.arange(half, dtype=np.float32) / np.float32(half)).astype(np.float32)[None]
        arg = gw.binary("Mul", "/time_proj", tf, gw.const("/time_proj", freqs))
        t = gw.op("Concat", "/time_proj", [gw.unary("Cos", "/time_proj", arg), gw.unary("Sin", "/time_proj", arg)], _attr_i("axis", -1))
    e = gw.gemm("/time_embedding/linear_1", "time_embedding.linear_1", w, t)
    e = gw.silu("/time_embedding/act", e)
    emb = gw.gemm("/time_embedding/linear_2", "time_embedding.linear_2", w, e)
    h = gw.conv("/conv_in", "conv_in", w, "sample")
    skips = [h]
    c = boc[0]
    for i, co in enumerate(boc):
        for j in range(lpb):
            h = _emit_resnet(gw, f"/down_blocks.{i}/resnets.{j}", f"down_blocks.{i}.resnets.{j}", w, h, emb, eps, groups)
            c = co
            if cfg["down_attn"][i]:
                h = _emit_transformer(gw, f"/down_blocks.{i}/attentions.{j}", f"down_blocks.{i}.attentions.{j}", w, h,
                                      "encoder_hidden_states", _heads(cfg, i), c, lin, groups)
            skips.append(h)
        if i < len(boc) - 1:
            h = gw.conv(f"/down_blocks.{i}/downsamplers.0/conv", f"down_blocks.{i}.downsamplers.0.conv", w, h, stride=2)
            skips.append(h)
    h = _emit_resnet(gw, "/mid_block/resnets.0", "mid_block.resnets.0", w, h, emb, eps, groups)
    h = _emit_transformer(gw, "/mid_block/attentions.0", "mid_block.attentions.0", w, h, "encoder_hidden_states",
                          _heads(cfg, len(boc) - 1), c, lin, groups)
    h = _emit_resnet(gw, "/mid_block/resnets.1", "mid_block.resnets.1", w, h, emb, eps, groups)
    up_attn = tuple(reversed(cfg["down_attn"]))
    for i, co in enumerate(reversed(boc)):
        for j in range(lpb + 1):
            s = skips.pop()
            cat = gw.tmp(f"/up_blocks.{i}", "Concat")
            gw.node("Concat", f"/up_blocks.{i}/Concat_{j}", [h, s], [cat], _attr_i("axis", 1))
            h = _emit_resnet(gw, f"/up_blocks.{i}/resnets.{j}", f"up_blocks.{i}.resnets.{j}", w, cat, emb, eps, groups)
            c = co
            if up_attn[i]:
                h = _emit_transformer(gw, f"/up_blocks.{i}/attentions.{j}", f"up_blocks.{i}.attentions.{j}", w, h,
                                      "encoder_hidden_states", _heads(cfg, len(boc) - 1 - i), c, lin, groups)
        if i < len(boc) - 1:
            sc = f"/up_blocks.{i}/upsamplers.0"
            r = gw.tmp(sc, "Resize")
            rattr = _attr_s("coordinate_transformation_mode", "asymmetric") + _attr_s("mode", "nearest") + _attr_s("nearest_mode", "floor")
            if fused:
                gw.node("Resize", sc + "/Resize", [h, "", gw.const(sc, np.asarray([1, 1, 2, 2], np.float32))], [r], rattr)
            else:
                # forward_upsample_size: F.interpolate(h, size = the next skip connection's H x W)
                nc = gw.op("Slice", sc, [gw.op("Shape", sc, [h]), gw.i64(sc, [0]), gw.i64(sc, [2]), gw.i64(sc, [0])])
                hw = gw.op("Slice", sc, [gw.op("Shape", sc, [skips[-1]]), gw.i64(sc, [2]), gw.i64(sc, [4]), gw.i64(sc, [0])], name="Slice_1")
                gw.node("Resize", sc + "/Resize", [h, "", "", gw.op("Concat", sc, [nc, hw], _attr_i("axis", 0))], [r], rattr)
            h = gw.conv(sc + "/conv", f"up_blocks.{i}.upsamplers.0.conv", w, r)
    h = gw.group_norm("/conv_norm_out", "conv_norm_out", w, h, eps, groups, swish=True)
    if not gw.fused:
        h = gw.silu("/conv_act", h)
    out = gw.conv("/conv_out", "conv_out", w, h)
    gw.node("Cast", "/Cast_out", [out], ["out_sample"], _attr_i("to", 1))
    nbytes = sum(a.nbytes for _, a in gw.inits)
    if external is None:
        external = nbytes > 2000 * 1024 * 1024  # sd2pyke.py:273-274 needs_collate
    gw.serialize(path, ["sample", "timestep", "encoder_hidden_states"], ["out_sample"], "unet_weights.bin" if external else None)


def write_vae_decoder_onnx(path: str, cfg: dict, w: Dict[str, np.ndarray], dtype=np.float32, fused: bool = False, nv: bool = False):
    gw = GraphWriter(dtype, fused, False, nv)
    groups = cfg["norm_num_groups"]
    boc = list(reversed(cfg["block_out_channels"]))
    z = gw.conv("/post_quant_conv", "post_quant_conv", w, "latent_sample", k=1)
    h = gw.conv("/decoder/conv_in", "decoder.conv_in", w, z)
    h = _emit_resnet(gw, "/decoder/mid_block/resnets.0", "decoder.mid_block.resnets.0", w, h, None, 1e-6, groups)
    sc, p = "/decoder/mid_block/attentions.0", "decoder.mid_block.attentions.0"
    c = boc[0]
    n = gw.group_norm(sc + "/group_norm", p + ".group_norm", w, h, 1e-6, groups)
    if gw.fused:
        # fusion_attention_vae.py:160-190: one Attention node, weight np.stack((q,k,v), axis=1); REFERENCE_QUIRK: the q/k/v biases
        # are stacked and then overwritten with zeros (:182-183), so the fused decoder runs without them
        qw, kw, vw = (np.ascontiguousarray(w[p + f".{t}.weight"].T) for t in ("query", "key", "value"))
        name = gw.fname("Attention")
        gw.init(name + "_qkv_weight", np.stack((qw, kw, vw), axis=1).reshape(c, 3 * c))
        gw.init(name + "_qkv_bias", np.zeros(3 * c, np.float32))
        o = gw.tmp(sc, "Reshape_3")
        gw.node("Attention", name, [n, name + "_qkv_weight", name + "_qkv_bias"], [o], _attr_i("num_heads", 1), "com.microsoft")
    else:
        q = gw.linear(sc + "/query", p + ".query", w, n)
        k = gw.linear(sc + "/key", p + ".key", w, n)
        v = gw.linear(sc + "/value", p + ".value", w, n)
        s = gw.binary("MatMul", sc, q, k)
        s = gw.binary("Mul", sc, s, gw.const(sc, np.asarray(c ** -0.5, np.float32)), "Mul_1")
        s = gw.unary("Softmax", sc, s)
        o = gw.binary("MatMul", sc, s, v, "MatMul_1")
    o = gw.linear(sc + "/proj_attn", p + ".proj_attn", w, o)
    h = gw.binary("Add", sc, o, h)
    h = _emit_resnet(gw, "/decoder/mid_block/resnets.1", "decoder.mid_block.resnets.1", w, h, None, 1e-6, groups)
    for i, co in enumerate(boc):
        for j in range(cfg["layers_per_block"] + 1):
            h = _emit_resnet(gw, f"/decoder/up_blocks.{i}/resnets.{j}", f"decoder.up_blocks.{i}.resnets.{j}", w, h, None, 1e-6, groups)
        if i < len(boc) - 1:
            s2 = f"/decoder/up_blocks.{i}/upsamplers.0"
            r = gw.tmp(s2, "Resize")
            gw.node("Resize", s2 + "/Resize", [h, "", gw.const(s2, np.asarray([1, 1, 2, 2], np.float32))], [r],
                    _attr_s("coordinate_transformation_mode", "asymmetric") + _attr_s("mode", "nearest") +
                    _attr_s("nearest_mode", "floor"))
            h = gw.conv(s2 + "/conv", f"decoder.up_blocks.{i}.upsamplers.0.conv", w, r)
    h = gw.group_norm("/decoder/conv_norm_out", "decoder.conv_norm_out", w, h, 1e-6, groups, swish=True)
    if not gw.fused:
        h = gw.silu("/decoder/conv_act", h)
    out = gw.conv("/decoder/conv_out", "decoder.conv_out", w, h)
    gw.node("Cast", "/Cast_out", [out], ["sample"], _attr_i("to", 1))
    gw.serialize(path, ["latent_sample"], ["sample"], None)


def write_vae_encoder_onnx(path: str, cfg: dict, w: Dict[str, np.ndarray], dtype=np.float32):
    """vae_encoder.onnx in the raw dialect (sd2pyke.py:181-187,325-346): initializer names and node kinds of the reference's
    fixture export -- Pad (0,1,0,1) + stride-2 pad-0 Conv downsamplers, Slice / Clip / Exp / RandomNormalLike posterior sample.
    The token reshapes around the mid-block attention are abbreviated as in write_vae_decoder_onnx (the engine and the oracle's
    restatement bind initializers by name)."""
    gw = GraphWriter(dtype, False, False, False)
    groups = cfg["norm_num_groups"]
    boc = list(cfg["block_out_channels"])
    lc = cfg["latent_channels"]
    h = gw.conv("/encoder/conv_in", "encoder.conv_in", w, "sample")
    for i, co in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            h = _emit_resnet(gw, f"/encoder/down_blocks.{i}/resnets.{j}", f"encoder.down_blocks.{i}.resnets.{j}", w, h, None, 1e-6, groups)
        if i < len(boc) - 1:
            s2 = f"/encoder/down_blocks.{i}/downsamplers.0"
            pd = gw.op("Pad", s2, [h, gw.i64(s2, [0, 0, 0, 0, 0, 0, 1, 1]), gw.const(s2, np.asarray(0.0, np.float32))], _attr_s("mode", "constant"))
            h = gw.conv(s2 + "/conv", f"encoder.down_blocks.{i}.downsamplers.0.conv", w, pd, stride=2, pad=0)
    h = _emit_resnet(gw, "/encoder/mid_block/resnets.0", "encoder.mid_block.resnets.0", w, h, None, 1e-6, groups)
    sc, p = "/encoder/mid_block/attentions.0", "encoder.mid_block.attentions.0"
    c = boc[-1]
    n = gw.group_norm(sc + "/group_norm", p + ".group_norm", w, h, 1e-6, groups)
    q = gw.linear(sc + "/query", p + ".query", w, n)
    k = gw.linear(sc + "/key", p + ".key", w, n)
    v = gw.linear(sc + "/value", p + ".value", w, n)
    s = gw.binary("MatMul", sc, q, k)
    s = gw.binary("Mul", sc, s, gw.const(sc, np.asarray(c ** -0.5, np.float32)), "Mul_1")
    s = gw.unary("Softmax", sc, s)
    o = gw.binary("MatMul", sc, s, v, "MatMul_1")
    o = gw.linear(sc + "/proj_attn", p + ".proj_attn", w, o)
    h = gw.binary("Add", sc, o, h)
    h = _emit_resnet(gw, "/encoder/mid_block/resnets.1", "encoder.mid_block.resnets.1", w, h, None, 1e-6, groups)
    h = gw.group_norm("/encoder/conv_norm_out", "encoder.conv_norm_out", w, h, 1e-6, groups, swish=True)
    h = gw.silu("/encoder/conv_act", h)
    h = gw.conv("/encoder/conv_out", "encoder.conv_out", w, h)
    m = gw.conv("/quant_conv", "quant_conv", w, h, k=1)
    ax = gw.i64("/post", [1])
    mean = gw.op("Slice", "/post", [m, gw.i64("/post", [0]), gw.i64("/post", [lc]), ax], name="Slice")
    logvar = gw.op("Slice", "/post", [m, gw.i64("/post", [lc]), gw.i64("/post", [2 * lc]), ax], name="Slice_1")
    logvar = gw.op("Clip", "/post", [logvar, gw.const("/post", np.asarray(-30.0, np.float32)), gw.const("/post", np.asarray(20.0, np.float32))])
    std = gw.unary("Exp", "/post", gw.binary("Mul", "/post", logvar, gw.const("/post", np.asarray(0.5, np.float32))))
    nz = gw.op("RandomNormalLike", "/post", [mean], _attr_i("dtype", 1))
    out = gw.binary("Add", "/post", mean, gw.binary("Mul", "/post", std, nz, "Mul_1"))
    gw.node("Cast", "/Cast_out", [out], ["latent_sample"], _attr_i("to", 1))
    gw.serialize(path, ["sample"], ["latent_sample"], None)


TOML_TEMPLATE = """v = 2
pipeline = "stable-diffusion"

[framework]
type = "orte"
opset = 15

[tokenizer]
type = "CLIPTokenizer"
path = "tokenizer.json"
model-max-length = 77
bos-token = 49406
eos-token = 49407

[text-encoder]
path = "text_encoder.onnx"

[text-encoder.text-embeddings]
path = "text_embeddings.bin"

[unet]
path = "unet.onnx"

[vae]
decoder = "vae_decoder.onnx"
scale-factor = 0.18215

[hashes]
text-encoder = "00000000000000000000000000000000"
text-embeddings = "00000000000000000000000000000000"
unet = "{unet_hash}"
vae-decoder = "{vae_hash}"
"""


def write_model_dir(root: str, unet_cfg: dict, vae_cfg: dict, seed: int = 42, dtype=np.float32,
                    unet_gain: float = 1.0, external: bool | None = None, fused: bool = False, approximate_gelu: bool = False,
                    nv: bool = False, vae_encoder: bool = False):
    """Writes pyke-diffusers.toml + unet.onnx(+unet_weights.bin) + vae_decoder.onnx (+ vae_encoder.onnx) with random weights.
    The text-encoder/tokenizer entries are placeholders: CLIP stays with the caller."""
    os.makedirs(root, exist_ok=True)
    uw = random_weights(unet_param_shapes(unet_cfg), seed, unet_gain)
    write_unet_onnx(os.path.join(root, "unet.onnx"), unet_cfg, uw, dtype, external, fused, approximate_gelu, nv)
    del uw
    vw = random_weights(vae_decoder_param_shapes(vae_cfg), seed + 1)
    write_vae_decoder_onnx(os.path.join(root, "vae_decoder.onnx"), vae_cfg, vw, dtype, fused, nv)
    toml = TOML_TEMPLATE.format(unet_hash=f"{seed:032x}", vae_hash=f"{seed + 1:032x}")
    if vae_encoder:
        ew = random_weights(vae_encoder_param_shapes(vae_cfg), seed + 2)
        write_vae_encoder_onnx(os.path.join(root, "vae_encoder.onnx"), vae_cfg, ew, dtype)
        toml = toml.replace('[vae]\ndecoder', '[vae]\nencoder = "vae_encoder.onnx"\ndecoder') + f'vae-encoder = "{seed + 2:032x}"\n'
    with open(os.path.join(root, "pyke-diffusers.toml"), "w") as f:
        f.write(toml)
    return root
