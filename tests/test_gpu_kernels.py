"""GPU: each hand-written MFMA kernel against (a) the plain on-device kernel and (b) a torch fp32 reference of the same
op on the fp16-rounded operands.  Tolerance: fp16 storage + fp32 accumulation => rel-L2 <= 2e-3 (written below)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def engine(tiny_model_dir):
    from diffusers_amd.pipeline import StableDiffusionPipeline
    p = StableDiffusionPipeline(tiny_model_dir)
    yield p
    p.close()


def run_conv(engine, naive, x, w, b, k, stride, ups):
    from diffusers_amd import _lib
    NB, Hi, Wi, Cin = x.shape
    Cout = w.shape[0]
    Hv, Wv = (2 * Hi - (ups == 2), 2 * Wi - (ups == 2)) if ups else (Hi, Wi)
    Ho, Wo = ((Hv - 1) // stride + 1, (Wv - 1) // stride + 1) if k == 3 else (Hv, Wv)
    out = np.empty((NB, Ho, Wo, Cout), np.float32)
    _lib.check(_lib.lib().sde_test_conv(engine._h, naive, x.ctypes.data, NB, Hi, Wi, Cin, w.ctypes.data,
                                        b.ctypes.data if b is not None else None, Cout, k, stride, int(ups), out.ctypes.data, out.size))
    return out


CONV_CASES = [
    # NB, H, W, Cin, Cout, k, stride, ups
    (2, 16, 16, 64, 160, 3, 1, 0),
    (1, 32, 32, 320, 320, 3, 1, 0),     # BM=128, BN=160
    (2, 16, 16, 128, 128, 3, 2, 0),     # stride-2 downsample, BN=128
    (1, 8, 8, 128, 256, 3, 1, 1),       # fused nearest-2x upsample
    (2, 9, 8, 128, 160, 3, 1, 2),       # fused upsample to 17 x 15 (skip-connection size on odd latents)
    (1, 13, 11, 64, 64, 3, 1, 0),       # ragged M (not a multiple of the tile), odd spatial size
    (1, 15, 9, 64, 64, 3, 2, 0),        # stride 2 on odd sizes
    (3, 8, 8, 192, 320, 1, 1, 0),       # 1x1 / linear
    (1, 64, 64, 64, 4, 3, 1, 0),        # conv_out-like: N=4 padded to one tile
    (1, 7, 5, 24, 40, 3, 1, 0),         # unaligned channels -> plain kernel on both paths
    (2, 8, 8, 1280, 1280, 3, 1, 0),     # few tiles, long K -> split-K partial slabs + reduce kernel
    (1, 16, 16, 640, 320, 3, 2, 0),     # split-K with stride 2
    (1, 24, 24, 2048, 192, 1, 1, 0),    # split-K on a linear layer, ragged M tile and padded N
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_mfma_vs_plain_vs_torch(engine, case):
    NB, H, W, Cin, Cout, k, stride, ups = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    fast = run_conv(engine, 0, x, w, b, k, stride, ups)
    plain = run_conv(engine, 1, x, w, b, k, stride, ups)
    xt = torch.from_numpy(x).half().float().permute(0, 3, 1, 2)
    if ups:
        xt = F.interpolate(xt, scale_factor=2.0, mode="nearest")
        if ups == 2:   # upsample to (2H-1, 2W-1): F.interpolate(size=...) nearest == the 2x result without its last row / column
            assert torch.equal(xt[:, :, :-1, :-1], F.interpolate(xt[:, :, ::2, ::2], size=(xt.shape[2] - 1, xt.shape[3] - 1), mode="nearest"))
            xt = xt[:, :, :-1, :-1]
    ref = F.conv2d(xt, torch.from_numpy(w).half().float(), torch.from_numpy(b), stride=stride if k == 3 else 1,
                   padding=1 if k == 3 else 0).permute(0, 2, 3, 1).numpy()
    assert fast.shape == ref.shape
    assert rel_l2(plain, ref) < 2e-3
    assert rel_l2(fast, ref) < 2e-3
    assert rel_l2(fast, plain) < 2e-3


BIG_CASES = [
    # NB, H, W, Cin, Cout, k, stride, ups  -- shapes the 8-wave 256x320 kernel accepts (Cout % 320 == 0, Cin % 64 == 0)
    (1, 32, 32, 320, 320, 3, 1, 0),
    (2, 16, 16, 128, 640, 3, 1, 0),     # two column tiles
    (1, 24, 20, 64, 320, 3, 1, 0),      # ragged M tile (480 rows), single K-tile per tap
    (1, 16, 16, 192, 320, 3, 2, 0),     # stride 2
    (1, 12, 12, 128, 320, 3, 1, 1),     # fused nearest-2x upsample
    (1, 30, 30, 256, 320, 1, 1, 0),     # linear, ragged M
    (1, 8, 8, 1280, 640, 3, 1, 0),      # long K (180 K-tiles): split-K slabs through the big kernel
    (1, 40, 40, 64, 320, 1, 1, 0),      # a single K-tile (prologue only)
    (1, 32, 32, 128, 256, 3, 1, 0),     # 256x256 tile variant (VAE widths)
    (1, 16, 16, 256, 512, 3, 1, 1),     # same, two column tiles, fused upsample
    (1, 17, 15, 320, 320, 3, 1, 2),     # fused upsample to 33 x 29 (odd latents), 256x320 tile
    (2, 24, 24, 128, 256, 1, 1, 0),     # same, linear, ragged M
]


@pytest.mark.parametrize("case", BIG_CASES)
def test_conv_big_tile_kernel(engine, case):
    NB, H, W, Cin, Cout, k, stride, ups = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    big = run_conv(engine, 2, x, w, b, k, stride, ups)
    xt = torch.from_numpy(x).half().float().permute(0, 3, 1, 2)
    if ups:
        xt = F.interpolate(xt, scale_factor=2.0, mode="nearest")
        if ups == 2:   # upsample to (2H-1, 2W-1): F.interpolate(size=...) nearest == the 2x result without its last row / column
            assert torch.equal(xt[:, :, :-1, :-1], F.interpolate(xt[:, :, ::2, ::2], size=(xt.shape[2] - 1, xt.shape[3] - 1), mode="nearest"))
            xt = xt[:, :, :-1, :-1]
    ref = F.conv2d(xt, torch.from_numpy(w).half().float(), torch.from_numpy(b), stride=stride if k == 3 else 1,
                   padding=1 if k == 3 else 0).permute(0, 2, 3, 1).numpy()
    assert big.shape == ref.shape
    assert rel_l2(big, ref) < 2e-3


HALO_CASES = [
    # NB, H, W, Cin, Cout, ups -- 3x3 stride-1 convs on the halo kernel (16 x 16 output patches, input staged once per 64-channel chunk)
    (1, 16, 16, 64, 160, 0),       # one patch, one chunk (prologue only: no second halo)
    (2, 32, 16, 128, 320, 0),      # two images x two patches, two chunks (the halo buffers swap once), two column tiles
    (1, 48, 32, 192, 160, 0),      # 3 x 2 patches: interior edges read the neighbour patch's pixels, outer edges the zero fill
    (1, 32, 32, 320, 320, 0),      # five chunks
    (3, 16, 16, 640, 640, 0),      # ten chunks, 4 column tiles
    (16, 16, 16, 1280, 1280, 0),   # the 16x16 UNet level at batch 16: 128 tiles, 2-way split-K at a chunk boundary
    (4, 16, 16, 1280, 160, 0),     # few tiles, 20 chunks: 8-way split (2-3 chunks per split)
    (2, 64, 64, 64, 160, 0),       # 32 patches, borders on every side
    # fused nearest-2x upsample (H, W = source size): a 10 x 10 source halo per patch
    (1, 8, 8, 64, 160, 1),         # one patch
    (2, 16, 8, 128, 320, 1),       # 2 x 1 patches per image, two chunks
    (1, 24, 16, 192, 160, 1),      # 3 x 2 patches, three chunks
    (16, 8, 8, 1280, 1280, 1),     # the 8x8 -> 16x16 upsampler at batch 16: split-K
    (2, 32, 32, 640, 640, 1),      # 32x32 -> 64x64, ten chunks
    # 256 x 128 tiles (widths that are multiples of 128 only: the VAE decoder)
    (1, 16, 16, 64, 128, 0),
    (2, 32, 48, 128, 128, 0),      # two chunks, 6 patches per image
    (1, 32, 32, 256, 256, 0),      # two column tiles
    (1, 16, 16, 512, 512, 1),      # fused upsample, four column tiles, eight chunks
    (1, 64, 64, 128, 384, 0),      # three column tiles
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_halo_kernel(engine, case):
    NB, H, W, Cin, Cout, ups = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = run_conv(engine, 5, x, w, b, 3, 1, ups)
    xt = torch.from_numpy(x).half().float().permute(0, 3, 1, 2)
    if ups:
        xt = F.interpolate(xt, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xt, torch.from_numpy(w).half().float(), torch.from_numpy(b), stride=1, padding=1).permute(0, 2, 3, 1).numpy()
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < 2e-3, rel_l2(got, ref)
    assert rel_l2(run_conv(engine, 3, x, w, b, 3, 1, ups), ref) < 2e-3      # the per-tap kernel it replaces, same operands


RES_CASES = [
    # kern, NB, H, W, Cin, Cout, k -- residual added in the epilogue, plain and as split-fp16 (hi + lo in, hi + lo out)
    (5, 2, 32, 16, 128, 320, 3),     # halo kernel: the epilogue maps tile rows to patch pixels for the residual reads too
    (5, 1, 48, 32, 192, 160, 3),
    (5, 16, 16, 16, 640, 1280, 3),   # 128 tiles, 2-way split-K: residual applied by the reduce pass
    (3, 1, 32, 32, 320, 320, 3),     # 256x160 per-tap kernel (loader waves)
    (2, 1, 32, 32, 320, 320, 3),     # 256x320
    (4, 4, 32, 32, 320, 640, 1),     # 128x160 linear kernel
    (0, 2, 16, 16, 128, 160, 1),     # whatever the heuristic picks for a small linear
    (1, 1, 8, 8, 64, 160, 3),        # plain kernel
]


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("case", RES_CASES)
def test_conv_residual_epilogue(engine, case, split):
    from diffusers_amd import _lib
    kern, NB, H, W, Cin, Cout, k = case
    rng = np.random.default_rng(hash(case) % (2 ** 31) + split)
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = (rng.standard_normal((NB, H, W, Cout)) * 3).astype(np.float32)
    out = np.empty((NB, H, W, Cout), np.float32)
    _lib.check(_lib.lib().sde_test_conv_res(engine._h, kern, x.ctypes.data, NB, H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, k,
                                            res.ctypes.data, split, out.ctypes.data, out.size))
    xt = torch.from_numpy(x).half().double().permute(0, 3, 1, 2)
    conv = F.conv2d(xt, torch.from_numpy(w).half().double(), torch.from_numpy(b).double(), stride=1, padding=k // 2).permute(0, 2, 3, 1).numpy()
    if split:
        # hi + lo carries ~22 bits: the residual goes in and the sum comes out essentially unrounded
        ref = conv + res.astype(np.float64)
        assert rel_l2(out, ref) < 2e-5, rel_l2(out, ref)
    else:
        ref = conv + torch.from_numpy(res).half().double().numpy()
        assert rel_l2(out, ref) < 5e-4, rel_l2(out, ref)


MID128_CASES = [
    # NB, H, W, Cin, Cout -- linears on the 128x160 loader-wave kernel (4-deep ring, loads three K-tiles ahead)
    (1, 16, 16, 64, 160),       # one K-tile: prologue only
    (1, 16, 16, 128, 320),      # two K-tiles
    (1, 20, 20, 192, 160),      # three K-tiles (the prologue fills the ring), ragged M (400 rows)
    (1, 16, 24, 256, 320),      # four: the first load issued inside the loop, into slot 3
    (2, 16, 16, 320, 480),      # five: the ring wraps
    (1, 30, 30, 384, 160),      # six, ragged M
    (16, 16, 16, 1280, 1280),   # the 16x16 UNet level at batch 16: 256 tiles, 20 K-tiles
    (4, 16, 16, 2560, 1280),    # 40 K-tiles
    (1, 24, 24, 704, 160),      # eleven K-tiles, ragged M (576 rows = 4.5 tiles)
]


@pytest.mark.parametrize("case", MID128_CASES)
def test_linear_mid128_kernel(engine, case):
    NB, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = run_conv(engine, 4, x, w, b, 1, 1, 0)
    xh = torch.from_numpy(x).half().float().reshape(-1, Cin)
    ref = (xh.double() @ torch.from_numpy(w).half().double().reshape(Cout, Cin).T + torch.from_numpy(b).double()).float().numpy().reshape(NB, H, W, Cout)
    assert rel_l2(got, ref) < 2e-3, rel_l2(got, ref)
    # the heuristic must agree with it whatever kernel it picks
    assert rel_l2(run_conv(engine, 0, x, w, b, 1, 1, 0), ref) < 2e-3


MID_CASES = [
    # NB, H, W, Cin, Cout, k, stride, ups -- shapes the 8-wave 256x160 kernel accepts (Cout % 160 == 0, Cin % 64 == 0)
    (1, 32, 32, 320, 320, 3, 1, 0),     # 1024 rows: 4 row tiles x 2 column tiles, 45 K-tiles (3-deep ring wraps many times)
    (2, 16, 16, 128, 640, 3, 1, 0),
    (1, 24, 20, 64, 160, 3, 1, 0),      # ragged M tile (480 rows), 9 K-tiles
    (1, 16, 16, 192, 320, 3, 2, 0),     # stride 2
    (1, 12, 12, 128, 480, 3, 1, 1),     # fused nearest-2x upsample, three column tiles
    (1, 9, 8, 128, 320, 3, 1, 2),       # fused upsample to 17 x 15 (odd latents)
    (1, 30, 30, 256, 320, 1, 1, 0),     # linear, ragged M, 4 K-tiles
    (1, 8, 8, 1280, 640, 3, 1, 0),      # long K: split-K slabs through this kernel (8 ways: <= 32 tiles)
    (16, 8, 8, 1280, 1280, 3, 1, 0),    # the 8x8 UNet level at UNet batch 16: 32 tiles x 8-way split-K = one block per CU
    (16, 16, 16, 640, 320, 3, 2, 0),    # stride-2 downsample into 8x8, 90 K-tiles, 8-way split
    (16, 32, 32, 640, 640, 3, 2, 0),    # stride-2 downsample into 16x16 at UNet batch 16: 64 tiles x 4-way split
    (1, 40, 40, 64, 160, 1, 1, 0),      # a single K-tile (prologue only)
    (1, 20, 20, 128, 320, 1, 1, 0),     # two K-tiles (prologue covers everything, loop issues nothing)
    (1, 20, 20, 192, 320, 1, 1, 0),     # three K-tiles (first use of the third ring slot)
]


@pytest.mark.parametrize("case", MID_CASES)
def test_conv_mid_tile_kernel(engine, case):
    NB, H, W, Cin, Cout, k, stride, ups = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    mid = run_conv(engine, 3, x, w, b, k, stride, ups)
    xt = torch.from_numpy(x).half().float().permute(0, 3, 1, 2)
    if ups:
        xt = F.interpolate(xt, scale_factor=2.0, mode="nearest")
        if ups == 2:   # upsample to (2H-1, 2W-1): F.interpolate(size=...) nearest == the 2x result without its last row / column
            assert torch.equal(xt[:, :, :-1, :-1], F.interpolate(xt[:, :, ::2, ::2], size=(xt.shape[2] - 1, xt.shape[3] - 1), mode="nearest"))
            xt = xt[:, :, :-1, :-1]
    ref = F.conv2d(xt, torch.from_numpy(w).half().float(), torch.from_numpy(b), stride=stride if k == 3 else 1,
                   padding=1 if k == 3 else 0).permute(0, 2, 3, 1).numpy()
    assert mid.shape == ref.shape
    assert rel_l2(mid, ref) < 2e-3


def run_attn(engine, naive, q, k, v, heads):
    from diffusers_amd import _lib
    NB, Sq, Cc = q.shape
    Skv = k.shape[1]
    out = np.empty_like(q)
    _lib.check(_lib.lib().sde_test_attention(engine._h, naive, q.ctypes.data, k.ctypes.data, v.ctypes.data, NB, heads, Sq, Skv,
                                             Cc // heads, out.ctypes.data, out.size))
    return out


ATTN_CASES = [
    # NB, heads, Sq, Skv, d
    (2, 8, 256, 256, 40),     # SD1.5 level-0 head dim
    (1, 8, 1024, 1024, 40),   # 64-row-per-wave, double-buffered variant
    (1, 2, 1100, 1100, 40),   # same, ragged query block and masked last key tile
    (1, 3, 1030, 320, 64),    # same, d = 64 (no spare V column: VALU row sums), odd tile count
    (2, 8, 256, 77, 40),      # cross attention, ragged key count
    (1, 8, 64, 64, 160),      # deepest level
    (2, 8, 200, 152, 80),     # ragged query tile + 2-chunk prompt
    (1, 5, 576, 227, 64),     # SD2.1 head dim, 3-chunk prompt
    (1, 4, 64, 64, 16),       # tiny test model
    (1, 1, 64, 64, 64),       # tiny VAE attention (single head)
    (1, 2, 33, 7, 12),        # unaligned head dim -> plain kernel
    (2, 8, 4096, 77, 40),     # cross attention at the 64x64 level: persistent blocks walking query blocks, K / V loaded once
    (1, 8, 1100, 152, 40),    # same, three key tiles (last ragged), ragged last query block
    (1, 3, 2048, 192, 40),    # same, exactly three full tiles
    (1, 8, 4096, 227, 40),    # four key tiles: back on the per-query-block form
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_mfma_vs_plain_vs_torch(engine, case):
    NB, heads, Sq, Skv, d = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    q = rng.standard_normal((NB, Sq, heads * d)).astype(np.float32)
    k = rng.standard_normal((NB, Skv, heads * d)).astype(np.float32)
    v = rng.standard_normal((NB, Skv, heads * d)).astype(np.float32)
    fast = run_attn(engine, 0, q, k, v, heads)
    plain = run_attn(engine, 1, q, k, v, heads)
    def sp(t, S):
        return torch.from_numpy(t).half().float().view(NB, S, heads, d).transpose(1, 2)
    att = torch.softmax(sp(q, Sq) @ sp(k, Skv).transpose(-1, -2) * d ** -0.5, dim=-1)
    ref = (att @ sp(v, Skv)).transpose(1, 2).reshape(NB, Sq, heads * d).numpy()
    assert rel_l2(plain, ref) < 2e-3
    assert rel_l2(fast, ref) < 3e-3   # P is rounded to fp16 before the PV product
    assert rel_l2(fast, plain) < 3e-3


def test_attention_softmax_rescale_branch(engine):
    """Force the online-softmax running max to jump at a later KV tile (one huge score in tile 3)."""
    NB, heads, S, d = 1, 1, 256, 64
    rng = np.random.default_rng(5)
    q = rng.standard_normal((NB, S, d)).astype(np.float32)
    k = rng.standard_normal((NB, S, d)).astype(np.float32)
    v = rng.standard_normal((NB, S, d)).astype(np.float32)
    k[0, 200] = q[0, 17] * 4.0   # spike: query 17 vs key 200 (4th tile)
    fast = run_attn(engine, 0, q, k, v, heads)
    qt, kt, vt = (torch.from_numpy(t).half().double() for t in (q, k, v))
    ref = (torch.softmax(qt @ kt.transpose(-1, -2) * d ** -0.5, dim=-1) @ vt).numpy()
    assert np.abs(fast - ref).max() < 2e-2 and rel_l2(fast, ref) < 3e-3


def test_hooks_refuse_a_buffer_of_the_wrong_size(engine):
    """The hooks compute their own output shape; a caller whose arithmetic disagrees (here: the symmetric-padding formula
    (H - 1) // 2 + 1 for the asymmetric-padding window on odd sizes, 8 x 5 instead of 7 x 4 rows) is refused before any
    copy -- the class of host heap overrun behind round 2's segfault in a GPU-test process."""
    from diffusers_amd import _lib
    L = _lib.lib()
    NB, H, W, Cin, Cout = 1, 15, 9, 64, 64
    x = np.zeros((NB, H, W, Cin), np.float32)
    w = np.zeros((Cout, Cin, 3, 3), np.float32)
    guard = np.full(8 * 5 * Cout + 64, 7.0, np.float32)
    for cap in (8 * 5 * Cout, 7 * 4 * Cout - 1, 0):
        st = L.sde_test_conv(engine._h, 0, x.ctypes.data, NB, H, W, Cin, w.ctypes.data, None, Cout, 3, 2, 3, guard.ctypes.data, cap)
        assert st == 1 and b"out_capacity" in L.sde_last_error()
    assert (guard == 7.0).all()
    st = L.sde_test_conv_res(engine._h, 0, x.ctypes.data, NB, H, W, Cin, w.ctypes.data, None, Cout, 3, guard.ctypes.data, 0, guard.ctypes.data, 5)
    assert st == 1 and b"out_capacity" in L.sde_last_error()
    q = np.zeros((1, 64, 64), np.float32)
    st = L.sde_test_attention(engine._h, 0, q.ctypes.data, q.ctypes.data, q.ctypes.data, 1, 4, 64, 64, 16, guard.ctypes.data, 64 * 64 + 1)
    assert st == 1 and b"out_capacity" in L.sde_last_error()
    assert (guard == 7.0).all()


# ---------------------------------------------------------------------------------------------- W-stationary streaming kernel (gemm_ws.hip)
WS_CASES = [
    # NB, H, W, Cin, Cout -- linears with K = 320 (160-column panels; anything else falls through to the tile kernels)
    (1, 16, 16, 320, 160),       # one chunk, one panel
    (1, 64, 32, 320, 320),       # 8 chunks (one per XCD), two panels side by side
    (2, 64, 64, 320, 320),       # 32 chunks
    (1, 48, 40, 320, 480),       # 1,920 rows = 7.5 chunks (ragged last chunk), three panels
    (1, 33, 31, 320, 160),       # 1,023 rows: ragged wave tile
    (4, 64, 64, 320, 960),       # the q / k / v projection shape of the 64x64 level (six panels: several passes)
    (1, 40, 40, 320, 1920),      # twelve panels, ragged rows (1,600 = 12.5 chunks of 128)
    (1, 40, 40, 640, 320),       # K = 640: not this kernel's (the hook falls through to the heuristic)
    (1, 7, 5, 320, 160),         # fewer rows than one wave tile pair
    (3, 50, 50, 320, 328),       # N not a multiple of the panel width (Npad = 480): masked columns
]


@pytest.mark.parametrize("case", WS_CASES)
def test_linear_ws_kernel(engine, case):
    NB, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    got = run_conv(engine, 6, x, w, b, 1, 1, 0)
    xh = torch.from_numpy(x).half().double().reshape(-1, Cin)
    ref = (xh @ torch.from_numpy(w).half().double().reshape(Cout, Cin).T + torch.from_numpy(b).double()).float().numpy().reshape(NB, H, W, Cout)
    assert rel_l2(got, ref) < 2e-3, rel_l2(got, ref)
    assert np.abs(got - ref).max() < 2e-2 * max(1.0, np.abs(ref).max())     # no stray rows / columns


WS_RES_CASES = [
    # NB, H, W, Cin, Cout -- residual in the epilogue, plain and split fp16 (the split-fp16 instance runs on 80-column panels)
    (1, 64, 32, 320, 320), (1, 40, 40, 320, 320), (2, 64, 64, 320, 640), (1, 33, 31, 320, 160), (1, 20, 20, 320, 168),
]


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("case", WS_RES_CASES)
def test_linear_ws_residual_epilogue(engine, case, split):
    from diffusers_amd import _lib
    NB, H, W, Cin, Cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31) + split)
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = (rng.standard_normal((NB, H, W, Cout)) * 3).astype(np.float32)
    out = np.empty((NB, H, W, Cout), np.float32)
    _lib.check(_lib.lib().sde_test_conv_res(engine._h, 6, x.ctypes.data, NB, H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, 1,
                                            res.ctypes.data, split, out.ctypes.data, out.size))
    lin = (torch.from_numpy(x).half().double().reshape(-1, Cin) @ torch.from_numpy(w).half().double().reshape(Cout, Cin).T
           + torch.from_numpy(b).double()).numpy().reshape(NB, H, W, Cout)
    if split:
        assert rel_l2(out, lin + res.astype(np.float64)) < 2e-5
    else:
        assert rel_l2(out, lin + torch.from_numpy(res).half().double().numpy()) < 5e-4


GEGLU_CASES = [
    # M, K, N, tanh -- N = value | gate rows; N % 320 == 0: rows interleaved in fours (256x320 kernel, W-stationary kernel)
    (1024, 64, 320, 0), (2048, 320, 2560, 0), (1600, 320, 640, 1), (4096, 640, 5120, 0), (1000, 640, 640, 0), (300, 128, 320, 1), (333, 320, 320, 0), (70, 640, 960, 1),
]


@pytest.mark.parametrize("kern", [0, 2, 6])
@pytest.mark.parametrize("case", GEGLU_CASES)
def test_geglu_projection(engine, case, kern):
    """The GEGLU projection with the gate product in the epilogue, on the heuristic's choice, the 256x320 kernel and the W-stationary kernel."""
    from diffusers_amd import _lib
    M, K, N, tanh = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = np.empty((M, N // 2), np.float32)
    _lib.check(_lib.lib().sde_test_geglu(engine._h, kern, x.ctypes.data, M, K, w.ctypes.data, b.ctypes.data, N, tanh, out.ctypes.data, out.size))
    y = torch.from_numpy(x).half().double() @ torch.from_numpy(w).half().double().T + torch.from_numpy(b).double()
    a, g = y[:, :N // 2], y[:, N // 2:]
    ref = (a * (F.gelu(g, approximate="tanh") if tanh else F.gelu(g))).numpy()
    assert rel_l2(out, ref) < 2e-3, rel_l2(out, ref)


# ---------------------------------------------------------------------------------------------- the fp32 engine's kernels (f32path.hip)
F32_CONV_CASES = CONV_CASES + [
    (2, 64, 64, 320, 320, 3, 1, 0),     # 128 x 128 tiles (>= 192 of them)
    (2, 32, 32, 640, 1280, 1, 1, 0),    # 64 x 64 tiles, linear
    (1, 10, 10, 48, 80, 3, 1, 0),       # 48 channels per tap: 16-float slabs inside a tap without the 64-channel chunking
    (1, 9, 9, 20, 33, 1, 1, 0),         # K % 16 != 0: plain kernel on both paths
]


@pytest.mark.parametrize("case", F32_CONV_CASES)
def test_f32_conv_mfma_vs_plain_vs_torch(engine, case):
    """naive 7 / 8 = the fp32 engine's MFMA-tile / plain FMA kernel on unrounded fp32 operands; reference in fp64.  Bar 1e-5 rel-L2
    (fp32 products and sums only; the engine-level bar is 1e-4)."""
    NB, H, W, Cin, Cout, k, stride, ups = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    fast = run_conv(engine, 7, x, w, b, k, stride, ups)
    plain = run_conv(engine, 8, x, w, b, k, stride, ups)
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    if ups:
        xt = F.interpolate(xt, scale_factor=2.0, mode="nearest")
        if ups == 2:
            xt = xt[:, :, :-1, :-1]
    ref = F.conv2d(xt, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride if k == 3 else 1,
                   padding=1 if k == 3 else 0).permute(0, 2, 3, 1).numpy()
    assert fast.shape == ref.shape
    assert rel_l2(plain, ref) < 1e-5
    assert rel_l2(fast, ref) < 1e-5


@pytest.mark.parametrize("case", ATTN_CASES + [(2, 8, 4096, 4096, 40), (1, 1, 300, 300, 512)])
def test_f32_attention_mfma_vs_plain_vs_torch(engine, case):
    NB, heads, Sq, Skv, d = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    q = rng.standard_normal((NB, Sq, heads * d)).astype(np.float32)
    k = rng.standard_normal((NB, Skv, heads * d)).astype(np.float32)
    v = rng.standard_normal((NB, Skv, heads * d)).astype(np.float32)
    fast = run_attn(engine, 7, q, k, v, heads)
    def sp(t, S):
        return torch.from_numpy(t).double().view(NB, S, heads, d).transpose(1, 2)
    att = torch.softmax(sp(q, Sq) @ sp(k, Skv).transpose(-1, -2) * d ** -0.5, dim=-1)
    ref = (att @ sp(v, Skv)).transpose(1, 2).reshape(NB, Sq, heads * d).numpy()
    assert rel_l2(fast, ref) < 1e-5
    if Sq * Skv <= 1 << 21:
        assert rel_l2(run_attn(engine, 8, q, k, v, heads), ref) < 1e-5


def test_f32_attention_softmax_rescale_branch(engine):
    NB, heads, S, d = 1, 1, 256, 64
    rng = np.random.default_rng(5)
    q = rng.standard_normal((NB, S, d)).astype(np.float32)
    k = rng.standard_normal((NB, S, d)).astype(np.float32)
    v = rng.standard_normal((NB, S, d)).astype(np.float32)
    k[0, 200] = q[0, 17] * 4.0
    fast = run_attn(engine, 7, q, k, v, heads)
    qt, kt, vt = (torch.from_numpy(t).double() for t in (q, k, v))
    ref = (torch.softmax(qt @ kt.transpose(-1, -2) * d ** -0.5, dim=-1) @ vt).numpy()
    assert rel_l2(fast, ref) < 1e-5 and np.abs(fast - ref).max() < 1e-4
