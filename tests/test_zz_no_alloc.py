"""GPU: sde_approximate_decode_noalloc (static device scratch, no hipMalloc) against sde_approximate_decode, through the C ABI.
Bar: bit-exact (same kernel on the same f32 inputs).  Two calls in a row, the second smaller (the array keeps the first call's bytes);
then a size beyond the static scratch, which must be refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hip_free_bytes_fn():
    """hipMemGetInfo of the HIP runtime the engine itself is linked to (the copy already mapped into this process: torch ships its own
    libamdhip64, and a second runtime initialised after the first finds no GPU -- so torch.cuda is not used here)."""
    import ctypes
    path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)
    hip = ctypes.CDLL(path)
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]

    def free_bytes():
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value
    return free_bytes


def test_noalloc_matches_allocating_entry_point(tiny_model_dir):
    from diffusers_amd import _lib
    from diffusers_amd.pipeline import StableDiffusionPipeline
    L = _lib.lib()
    pipe = StableDiffusionPipeline(tiny_model_dir)
    try:
        free_bytes = _hip_free_bytes_fn()
        rng = np.random.default_rng(70)
        warm = rng.standard_normal((16, 4, 64, 64)).astype(np.float32)       # the allocating sibling grows its buffers to the largest case here,
        pipe.approximate_decode_latents(warm)                                 # and one no-alloc call pins whatever the runtime sets up lazily
        warm_out = np.empty((16, 64, 64, 3), np.float32)                      # kept in a name: a temporary's buffer would be freed before the call
        _lib.check(L.sde_approximate_decode_noalloc(pipe._h, warm.ctypes.data, 16, 64, 64, warm_out.ctypes.data))
        for B, h, w in [(8, 64, 64), (2, 9, 7), (1, 1, 1), (16, 64, 64)]:
            lat = rng.standard_normal((B, 4, h, w)).astype(np.float32)
            want = pipe.approximate_decode_latents(lat)
            got = np.full((B, h, w, 3), -7.0, np.float32)
            free0 = free_bytes()
            _lib.check(L.sde_approximate_decode_noalloc(pipe._h, lat.ctypes.data, B, h, w, got.ctypes.data))
            assert free_bytes() == free0, "the no-allocation entry point changed the device's free memory"
            assert np.array_equal(got, want), (B, h, w)
            assert got.min() >= 0.0 and got.max() <= 1.0
        # 8 MiB of scratch: 16*n + 12*n bytes for n pixels -> n = 300,000 does not fit
        big = np.zeros((1, 4, 500, 600), np.float32)
        out = np.zeros((1, 500, 600, 3), np.float32)
        assert L.sde_approximate_decode_noalloc(pipe._h, big.ctypes.data, 1, 500, 600, out.ctypes.data) == 1
        assert b"static scratch" in L.sde_last_error()
        assert np.isfinite(pipe.approximate_decode_latents(big)).all()      # the allocating entry point takes it
    finally:
        pipe.close()
