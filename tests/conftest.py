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
