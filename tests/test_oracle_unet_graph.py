"""CPU: two routes from the same weights to the UNet output.

  route 1  oracle/unet_ref.py: functional torch restatement, weights bound by name (oracle/model_dir.py)
  route 2  the executable raw-dialect ONNX graph (diffusers_amd/modelgen.py) run node by node through oracle/onnx_interp.py,
           whose operator semantics are pinned on the reference's own torch-1.13 exports (tests/test_oracle_fixture_graphs.py)

The graph spells out what route 1 leaves to torch: which half of the GEGLU projection is the gate, cos-before-sin in the time
projection, the timestep's truncation, hidden-before-skip in the decoder concat, the head split / merge transposes, the softmax
scale, the upsample to the skip connection's size.  A misreading in one route that the other does not share shows up here; fp32, so
the bar is round-off."""
import numpy as np
import pytest
import torch

from diffusers_amd import modelgen
from oracle import model_dir, onnx_interp, onnx_io, unet_ref


def _routes(tmp_path, lin, heads, cases):
    ucfg = modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False), cross_attention_dim=64,
                                num_heads=heads, use_linear_projection=lin)
    vcfg = modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1)
    modelgen.write_model_dir(str(tmp_path), ucfg, vcfg, seed=11)
    uw, oc = model_dir.load_unet(str(tmp_path))
    uw = {k: torch.from_numpy(v) for k, v in uw.items()}
    m = onnx_io.load(str(tmp_path / "unet.onnx"))
    rng = np.random.default_rng(5)
    for N, h, w, S, t in cases:
        x = rng.standard_normal((N, 4, h, w)).astype(np.float32)
        e = (rng.standard_normal((N, S, 64)) * 0.5).astype(np.float32)
        got = onnx_interp.run(m, {"sample": x, "timestep": np.asarray([t], np.float32), "encoder_hidden_states": e})["out_sample"]
        with torch.no_grad():
            ref = unet_ref.unet_forward(uw, oc, torch.from_numpy(x), t, torch.from_numpy(e)).numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), (lin, N, h, w, S, t)
    return m, uw, oc


def test_sd1_style_graph_matches_restatement(tmp_path):
    m, uw, oc = _routes(tmp_path, False, 4, [(2, 16, 16, 77, 946.42), (1, 15, 17, 152, 3.0), (1, 8, 8, 227, 999.0)])
    # the graph really carries the semantics: truncation of the timestep ...
    x = np.random.default_rng(1).standard_normal((1, 4, 8, 8)).astype(np.float32)
    e = np.zeros((1, 77, 64), np.float32)
    f = lambda t: onnx_interp.run(m, {"sample": x, "timestep": np.asarray([t], np.float32), "encoder_hidden_states": e})["out_sample"]
    assert np.array_equal(f(946.42), f(946.0)) and not np.array_equal(f(946.42), f(945.0))
    # ... and a route-1 misreading would be caught: swapping the GEGLU halves in the restatement breaks the agreement
    k = "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj"
    bad = dict(uw)
    half = bad[k + ".weight"].shape[0] // 2
    bad[k + ".weight"] = torch.cat([uw[k + ".weight"][half:], uw[k + ".weight"][:half]])
    bad[k + ".bias"] = torch.cat([uw[k + ".bias"][half:], uw[k + ".bias"][:half]])
    with torch.no_grad():
        wrong = unet_ref.unet_forward(bad, oc, torch.from_numpy(x), 946.0, torch.from_numpy(e)).numpy()
    assert np.abs(wrong - f(946.0)).max() > 1e-3


def test_sd2_style_graph_matches_restatement(tmp_path):
    """Linear proj_in / proj_out (use_linear_projection) and per-level head counts."""
    _routes(tmp_path, True, (2, 4), [(2, 16, 16, 77, 500.0), (1, 9, 11, 77, 20.0)])
