"""GPU: short runs of the randomised parity sweeps (tools/fuzz_kernels.py, tools/fuzz_models.py): random conv / GEMM / attention
shapes through the C ABI test hooks against torch fp32, and random small UNet / VAE architectures (raw and optimizer-fused
dialects, odd latent sizes) against the CPU oracle.  The long sweeps are run by hand (round 2: 2,840 kernel cases, 112 models and 78
whole-loop runs; round 3, with the W-stationary kernel, the GEGLU / residual hooks and the fp32 engine's MFMA kernels in the draw:
1,400 kernel cases, 40 + 25 (fp32 engine) models, 30 whole-loop runs; no miss)."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _run(script, *args):
    argv = sys.argv
    sys.argv = [script, *map(str, args)]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(TOOLS, script), run_name="__main__")
        return e.value.code
    finally:
        sys.argv = argv


@pytest.mark.parametrize("seed", [11, 12])
def test_random_kernel_shapes(seed):
    assert _run("fuzz_kernels.py", 90, seed) == 0


def test_random_architectures():
    assert _run("fuzz_models.py", 6, 21) == 0


def test_degenerate_groupnorm_draw_is_pinned(tmp_path):
    """The fp16 engine's documented limit (sdengine.h): a GroupNorm group that holds only a handful of values -- here a
    32-channel level (ONE channel per group) whose deepest level is 1 x 1 -- normalises fp16 rounding noise by rstd ~ eps^-1/2, and the
    1e-2 bar does not hold; the fp32 engine meets its own bar on the same draw.  The fuzzer (tools/fuzz_models.py) keeps >= 8 values per
    group; this case records what happens below that, so that a regression OR an improvement shows up here."""
    import numpy as np
    import torch
    from diffusers_amd import modelgen
    from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline
    from oracle import model_dir, unet_ref
    d = str(tmp_path / "degenerate")
    ucfg = modelgen.unet_config(block_out_channels=(32, 32), layers_per_block=1, down_attn=(True, False), cross_attention_dim=32, num_heads=1)
    modelgen.write_model_dir(d, ucfg, modelgen.vae_config(block_out_channels=(32,), layers_per_block=1), seed=5)
    uw, oc = model_dir.load_unet(d)
    uw = {k: torch.from_numpy(v.astype(np.float32)) for k, v in uw.items()}
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 4, 2, 2)).astype(np.float32)            # 16 x 16 pixel "image": levels 2 x 2 and 1 x 1
    ehs = (rng.standard_normal((2, 9, 32)) * 0.5).astype(np.float32)
    with torch.no_grad():
        ref = unet_ref.unet_forward(uw, oc, torch.from_numpy(x), 500.0, torch.from_numpy(ehs)).numpy()
    errs = {}
    for dt in ("f32", "f16"):
        pipe = StableDiffusionPipeline(d, StableDiffusionOptions(dtype=dt), load_vae=False)
        try:
            got = pipe.unet(x, 500.0, ehs)
        finally:
            pipe.close()
        assert np.isfinite(got).all()
        errs[dt] = float(np.linalg.norm(got.astype(np.float64) - ref) / np.linalg.norm(ref))
    if os.environ.get("SDE_PARITY_LOG"):
        with open(os.environ["SDE_PARITY_LOG"], "a") as f:
            f.write(f"degenerate GroupNorm draw: f32 engine {errs['f32']:.3e}, f16 engine {errs['f16']:.3e}\n")
    # measured on MI355X (round 3): f32 engine 4.3e-4, f16 engine 0.54.  The draw is ill-conditioned in itself (four nearly equal values
    # per group at the 2 x 2 level: even two fp32 evaluation orders differ at 1e-4); both engines stay finite, and OUTSIDE their bars
    assert errs["f32"] < 2e-3, errs
    assert errs["f16"] < 1.5, errs
