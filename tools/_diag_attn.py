import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusers_amd import _lib, modelgen
from diffusers_amd.pipeline import StableDiffusionPipeline
d_ = "/tmp/mb_tiny"
if not os.path.exists(d_ + "/unet.onnx"):
    modelgen.write_model_dir(d_, modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False), cross_attention_dim=64, num_heads=4),
                             modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1), seed=1)
p = StableDiffusionPipeline(d_)
L = _lib.lib()
def run(naive, q, k, v, heads):
    NB, Sq, C = q.shape
    out = np.empty_like(q)
    _lib.check(L.sde_test_attention(p._h, naive, q.ctypes.data, k.ctypes.data, v.ctypes.data, NB, heads, Sq, k.shape[1], C // heads, out.ctypes.data))
    return out
for (NB, heads, Sq, Skv, d) in [(1, 1, 64, 64, 40), (1, 1, 64, 128, 40), (1, 1, 64, 256, 40), (1, 1, 32, 256, 64)]:
    rng = np.random.default_rng(1)
    q = rng.standard_normal((NB, Sq, heads * d)).astype(np.float32)
    k = rng.standard_normal((NB, Skv, heads * d)).astype(np.float32)
    v = rng.standard_normal((NB, Skv, heads * d)).astype(np.float32)
    f, pl = run(0, q, k, v, heads), run(1, q, k, v, heads)
    bad = ~(np.abs(f - pl) < 1e-2)
    print((NB, heads, Sq, Skv, d), "bad elements", int(bad.sum()), "of", bad.size, "nan", int(np.isnan(f).sum()))
    if bad.any():
        rows = np.where(bad[0].any(axis=1))[0]; cols = np.where(bad[0].any(axis=0))[0]
        print("  rows", rows[:40], "...", len(rows)); print("  cols", cols[:64], "...", len(cols))
        r0 = rows[0]; print("  row", r0, "fast", f[0, r0, :8], "plain", pl[0, r0, :8], "ratio", (f[0, r0, :8] / pl[0, r0, :8]))
p.close()
