"""CPU: pins the ONNX interpreter (oracle/onnx_interp.py) on graphs this repo did NOT write -- the reference's own fixture exports
tests/stable-diffusion/{text_encoder,vae_encoder}.onnx (torch 1.13.1, opset 15; copied as data to tests/golden/) -- against
independent functional torch restatements that share nothing with the interpreter but the initializers
(oracle/fixture_models.py).  Agreement to f32 round-off on real exporter patterns (decomposed LayerNorm, quick-GELU, masked
multi-head attention through Reshape / Transpose / Shape / Gather plumbing, Pad + stride-2 Conv, the Slice / Clip / Exp /
RandomNormalLike posterior) is what lets the interpreter serve as the second opinion on oracle/unet_ref.py
(tests/test_oracle_unet_graph.py)."""
import os

import numpy as np
import torch

from oracle import fixture_models as FM, onnx_interp, onnx_io

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_text_encoder_fixture_interpreter_vs_restatement():
    m = onnx_io.load(os.path.join(GOLD, "text_encoder.onnx"))
    g = np.load(os.path.join(GOLD, "text_encoder_io.npz"))
    out = onnx_interp.run(m, {"input_ids": g["input_ids"]})
    assert np.array_equal(out["last_hidden_state"], g["last_hidden_state"]) or np.abs(out["last_hidden_state"] - g["last_hidden_state"]).max() < 1e-6
    w = FM.bind_all(m)
    with torch.no_grad():
        h, pooled = FM.clip_text_encoder(w, torch.from_numpy(g["input_ids"]), heads=4)   # hidden 32 = 4 heads x 8
    assert h.shape == (2, 77, 32)
    assert np.abs(h.numpy() - out["last_hidden_state"]).max() < 5e-6
    assert np.abs(pooled.numpy() - out["pooler_output"]).max() < 5e-6
    # a wrong head count does not come close: the agreement above is not vacuous
    with torch.no_grad():
        h2, _ = FM.clip_text_encoder(w, torch.from_numpy(g["input_ids"]), heads=2)
    assert np.abs(h2.numpy() - out["last_hidden_state"]).max() > 1e-2


def test_vae_encoder_fixture_interpreter_vs_restatement():
    m = onnx_io.load(os.path.join(GOLD, "vae_encoder.onnx"))
    g = np.load(os.path.join(GOLD, "vae_encoder_io.npz"))
    nz = g["noise"]
    out = onnx_interp.run(m, {"sample": g["sample"]}, random_normal=lambda shp: nz.reshape(shp))["latent_sample"]
    assert np.abs(out - g["latent_sample"]).max() < 1e-6
    w = FM.bind_all(m)
    with torch.no_grad():
        ref = FM.vae_encoder(w, torch.from_numpy(g["sample"]), torch.from_numpy(nz)).numpy()
        mean = FM.vae_encoder(w, torch.from_numpy(g["sample"]), None).numpy()
    assert ref.shape == out.shape == (2, 4, 32, 24)
    assert np.abs(ref - out).max() < 5e-6 * max(1.0, np.abs(ref).max())
    # RandomNormalLike defaults to zeros in the interpreter: then the graph returns the posterior mean
    out0 = onnx_interp.run(m, {"sample": g["sample"]})["latent_sample"]
    assert np.abs(mean - out0).max() < 5e-6 * max(1.0, np.abs(mean).max())
