import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        from diffusers_amd import _lib
        return _lib.lib().sde_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests fail loudly (not skip) on a box without the extension; on a box without a GPU they are deselected by -m "not gpu"
    pass


@pytest.fixture(scope="session")
def tiny_model_dir(tmp_path_factory):
    """Synthetic tiny SD model dir (aligned channel counts so the MFMA kernels are exercised)."""
    from diffusers_amd import modelgen
    d = tmp_path_factory.mktemp("tiny_sd")
    ucfg = modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False),
                                cross_attention_dim=64, num_heads=4)
    vcfg = modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1)
    modelgen.write_model_dir(str(d), ucfg, vcfg, seed=7)
    return str(d)


def _tiny_cfgs():
    from diffusers_amd import modelgen
    ucfg = modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False),
                                cross_attention_dim=64, num_heads=4)
    vcfg = modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1)
    return ucfg, vcfg


@pytest.fixture(scope="session")
def tiny_model_dir_fused(tmp_path_factory):
    """Same weights as tiny_model_dir (seed 7) written in the optimizer-fused dialect (scripts/optimize.py --preset gpu without
    --approximate-gelu, before the fp16 conversion): com.microsoft Attention / MultiHeadAttention + MatMul_KV / GroupNorm,
    LayerNormalization."""
    from diffusers_amd import modelgen
    d = tmp_path_factory.mktemp("tiny_sd_fused")
    ucfg, vcfg = _tiny_cfgs()
    modelgen.write_model_dir(str(d), ucfg, vcfg, seed=7, fused=True)
    return str(d)


@pytest.fixture(scope="session")
def tiny_model_dir_fused_f16(tmp_path_factory):
    """The fused dialect after convert_float_to_float16: every initializer (biases and norm scales included) is fp16."""
    import numpy as np
    from diffusers_amd import modelgen
    d = tmp_path_factory.mktemp("tiny_sd_fused_f16")
    ucfg, vcfg = _tiny_cfgs()
    modelgen.write_model_dir(str(d), ucfg, vcfg, seed=7, dtype=np.float16, fused=True)
    return str(d)


@pytest.fixture(scope="session")
def tiny_model_dir_fused_nv(tmp_path_factory):
    """The `*-nv` presets (scripts/optimize.py:102-130): on top of the fused dialect, NhwcConv behind activation / weight Transposes,
    BiasSplitGelu, BiasAdd and SkipLayerNormalization."""
    from diffusers_amd import modelgen
    d = tmp_path_factory.mktemp("tiny_sd_fused_nv")
    ucfg, vcfg = _tiny_cfgs()
    modelgen.write_model_dir(str(d), ucfg, vcfg, seed=7, fused=True, nv=True)
    return str(d)


@pytest.fixture(scope="session")
def tiny_model_dir_fused_fastgelu(tmp_path_factory):
    """As above with --approximate-gelu: the feed-forward GELU is com.microsoft.FastGelu (tanh form)."""
    from diffusers_amd import modelgen
    d = tmp_path_factory.mktemp("tiny_sd_fused_fg")
    ucfg, vcfg = _tiny_cfgs()
    modelgen.write_model_dir(str(d), ucfg, vcfg, seed=7, fused=True, approximate_gelu=True)
    return str(d)
