"""CPU: the shared library loads and exports every symbol of include/sdengine.h; model-dir writer/reader round trip;
oracle VAE restatement pinned against the reference's fixture graph (committed under tests/golden/)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_header_symbols_exported():
    from diffusers_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sdengine.h")).read()
    declared = set(re.findall(r"^(?:sde_status|const char\*|int|float|void)\s+(sde_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert len(declared) >= 30
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in sdengine.h but not exported"
    assert declared <= set(_lib.SYMBOLS), declared - set(_lib.SYMBOLS)
    assert b"gfx950" in L.sde_version()


def test_engine_create_errors_without_model(tmp_path):
    import ctypes as C
    from diffusers_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    st = L.sde_engine_create(str(tmp_path).encode(), None, C.byref(h))
    assert st == 2 and b"pyke-diffusers.toml" in L.sde_last_error()            # SDE_ERR_IO
    (tmp_path / "pyke-diffusers.toml").write_text('pipeline = "stable-diffusion"\n[framework]\ntype = "tensorrt"\n')
    st = L.sde_engine_create(str(tmp_path).encode(), None, C.byref(h))
    assert st == 3 and b"bad framework" in L.sde_last_error()                  # impl_main.rs:113-118


def test_model_dir_round_trip(tiny_model_dir):
    from diffusers_amd import modelgen
    from oracle import model_dir
    uw, ucfg = model_dir.load_unet(tiny_model_dir)
    assert ucfg["block_out_channels"] == (64, 128) and ucfg["num_heads"] == 4 and ucfg["cross_attention_dim"] == 64
    ref = modelgen.random_weights(modelgen.unet_param_shapes(modelgen.unet_config(
        block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False), cross_attention_dim=64, num_heads=4)), 7)
    assert set(ref) == set(uw)
    assert all(np.array_equal(ref[k], uw[k]) for k in ref)
    vw, vcfg = model_dir.load_vae_decoder(tiny_model_dir)
    assert vcfg["block_out_channels"] == (64, 128)


def test_oracle_vae_matches_reference_fixture_graph():
    """tests/golden/vae_decoder.onnx is the reference's own fixture (tests/stable-diffusion/vae_decoder.onnx);
    golden I/O was produced by oracle/onnx_interp.py executing that graph (tests/golden/make_golden.py)."""
    from oracle import model_dir, onnx_io, onnx_interp, unet_ref
    m = onnx_io.load(os.path.join(GOLD, "vae_decoder.onnx"))
    g = np.load(os.path.join(GOLD, "vae_fixture_io.npz"))
    out = onnx_interp.run(m, {"latent_sample": g["latent"]})["sample"]
    assert np.allclose(out, g["sample"], atol=1e-5)
    w = model_dir.bind_weights(m)
    cfg = model_dir.infer_vae_config(w)
    assert cfg["block_out_channels"] == (32, 64) and cfg["layers_per_block"] == 1
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    got = unet_ref.vae_decode(tw, cfg, torch.from_numpy(g["latent"])).numpy()
    assert np.abs(got - g["sample"]).max() < 2e-5


def test_fused_dialect_writer_follows_the_optimizer_packing(tmp_path):
    """modelgen's fused dialect restates scripts/_optimizer/fusion_attention_unet.py:215-226 (Attention weight = np.stack((q,k,v),
    axis=1)), :231-280 (MatMul_KV = per-head interleave, Reshape [0,0,n,2,h]) and fusion_group_norm.py:108-160 (gamma/beta
    initializers named after the node).  Read the file back with the independent oracle reader and undo the packing."""
    import numpy as np
    from diffusers_amd import modelgen
    from oracle import onnx_io
    ucfg = modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False), cross_attention_dim=64, num_heads=4)
    w = modelgen.random_weights(modelgen.unet_param_shapes(ucfg), 3)
    path = str(tmp_path / "unet.onnx")
    modelgen.write_unet_onnx(path, ucfg, w, fused=True, approximate_gelu=True)
    m = onnx_io.load(path)
    ops = [n.op_type for n in m.nodes]
    assert "InstanceNormalization" not in ops and "Softmax" not in ops and "Erf" not in ops
    assert {"Attention", "MultiHeadAttention", "GroupNorm", "LayerNormalization", "FastGelu"} <= set(ops)
    att = [n for n in m.nodes if n.op_type == "Attention"][0]
    assert att.domain == "com.microsoft" and att.attr_i("num_heads") == 4
    p = "down_blocks.0.attentions.0.transformer_blocks.0"
    W = m.weight(att.inputs[1])
    C = W.shape[0]
    for j, t in enumerate("qkv"):
        assert np.array_equal(W[:, j * C:(j + 1) * C], w[f"{p}.attn1.to_{t}.weight"].T)
    kv = [n for n in m.nodes if n.name.startswith("MatMul_KV")][0]
    Wkv = m.weight(kv.inputs[1])
    heads, two, d = m.weight(kv.name + "_reshape_shape")[2:]
    assert (heads, two, d) == (4, 2, 16)
    Wkv = Wkv.reshape(Wkv.shape[0], heads, 2, d)
    assert np.array_equal(Wkv[:, :, 0, :].reshape(-1, heads * d), w[f"{p}.attn2.to_k.weight"].T)
    assert np.array_equal(Wkv[:, :, 1, :].reshape(-1, heads * d), w[f"{p}.attn2.to_v.weight"].T)
    gn = [n for n in m.nodes if n.op_type == "GroupNorm"][0]
    assert gn.inputs[1] == gn.name + "_gamma" and gn.attr_i("activation") == 1 and gn.attr_i("groups") == 32
    assert np.array_equal(m.weight(gn.inputs[1]), w["down_blocks.0.resnets.0.norm1.weight"])


def test_rust_sys_crate_declares_the_whole_abi():
    """rust/sdengine-sys/src/lib.rs (source only: no Rust toolchain here) must declare exactly the entry points of
    include/sdengine.h, and its struct mirrors must list the header's fields in the header's order."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "sdengine.h")).read()
    rs = open(os.path.join(root, "rust", "sdengine-sys", "src", "lib.rs")).read()
    c_fns = set(re.findall(r"\b(sde_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    c_fns -= {"sde_callback_fn", "sde_exchange_fn", "sde_exchange_stream_fn"}
    rs_fns = set(re.findall(r"pub fn (sde_[a-z0-9_]+)\s*\(", rs))
    assert c_fns == rs_fns, (sorted(c_fns - rs_fns), sorted(rs_fns - c_fns))
    from diffusers_amd import _lib
    assert rs_fns == set(_lib.SYMBOLS)

    def c_fields(name):
        body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", hdr, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                out.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1].lstrip("*")))
        return out

    def rs_fields(name):
        body = re.search(r"pub struct " + name + r" \{(.*?)\}\n", rs, flags=re.S).group(1)
        return re.findall(r"pub ([a-z_0-9]+):", body)

    for st in ("sde_engine_opts", "sde_sched_cfg", "sde_run_stats", "sde_model_info", "sde_kernel_stat", "sde_txt2img_opts"):
        cf, rf = [f.lower() for f in c_fields(st)], rs_fields(st)
        assert cf == rf, (st, cf, rf)
