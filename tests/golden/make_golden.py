"""Generates the committed golden fixtures.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

* vae_decoder.onnx      — copy of the reference's own test fixture data file
                          (/root/reference/tests/stable-diffusion/vae_decoder.onnx, a tiny random VAE exported by
                          torch 1.13.1; it is a data file the reference's tests hold, not source).
* vae_fixture_io.npz    — seeded latent + the output of that graph executed by oracle/onnx_interp.py
                          (ONNX Runtime is not available here, so the graph is interpreted op by op).
* text_encoder.onnx, vae_encoder.onnx
                        — copies of the reference's two other fixture data files (same directory): real torch-1.13 exports used
                          to pin the operator semantics of oracle/onnx_interp.py (tests/test_oracle_fixture_graphs.py).
* text_encoder_io.npz, vae_encoder_io.npz
                        — seeded inputs + the interpreter's outputs on those graphs (regression pins; the independent check is
                          the torch restatement in oracle/fixture_models.py).
* sched_traj_*.npz      — 3-step latent trajectories of the oracle's scheduler restatement on 1x4x8x8 latents
                          with fixed injected eps / noise (regression pins for oracle/schedulers.py).
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import onnx_io, onnx_interp, schedulers  # noqa: E402

REF = "/root/reference/tests/stable-diffusion/vae_decoder.onnx"


def main():
    shutil.copyfile(REF, os.path.join(HERE, "vae_decoder.onnx"))
    m = onnx_io.load(os.path.join(HERE, "vae_decoder.onnx"))
    rng = np.random.default_rng(1234)
    z = rng.standard_normal((2, 4, 8, 8)).astype(np.float32)
    out = onnx_interp.run(m, {"latent_sample": z})["sample"]
    np.savez_compressed(os.path.join(HERE, "vae_fixture_io.npz"), latent=z, sample=out)
    refdir = os.path.dirname(REF)
    for f in ("text_encoder.onnx", "vae_encoder.onnx"):
        shutil.copyfile(os.path.join(refdir, f), os.path.join(HERE, f))
    rng = np.random.default_rng(4321)
    ids = np.ones((2, 77), np.int32)                    # eos = 1 padding, bos = 0 (tests/stable-diffusion/tokenizer.json)
    ids[:, 0] = 0
    ids[0, 1:9] = rng.integers(2, 1000, 8); ids[1, 1:30] = rng.integers(2, 1000, 29)
    out = onnx_interp.run(onnx_io.load(os.path.join(HERE, "text_encoder.onnx")), {"input_ids": ids})
    np.savez_compressed(os.path.join(HERE, "text_encoder_io.npz"), input_ids=ids, **out)
    x = rng.standard_normal((2, 3, 64, 48)).astype(np.float32)
    nz = rng.standard_normal((2, 4, 32, 24)).astype(np.float32)
    out = onnx_interp.run(onnx_io.load(os.path.join(HERE, "vae_encoder.onnx")), {"sample": x}, random_normal=lambda shp: nz.reshape(shp))
    np.savez_compressed(os.path.join(HERE, "vae_encoder_io.npz"), sample=x, noise=nz, **out)
    for kind in schedulers.KINDS:
        for n in (20, 30):
            s = schedulers.make(kind)
            s.set_timesteps(n)
            r = np.random.default_rng(99)
            lat = (r.standard_normal((1, 4, 8, 8)).astype(np.float32) * s.init_noise_sigma()).astype(np.float32)
            eps = r.standard_normal((3, 1, 4, 8, 8)).astype(np.float32)
            noise = r.standard_normal((3, 1, 4, 8, 8)).astype(np.float32)
            traj = []
            x = lat
            for i in range(3):
                t = s.timesteps[i]
                s.scale_model_input(x, t)
                x = s.step(eps[i], t, x, noise[i])
                traj.append(x)
            np.savez_compressed(os.path.join(HERE, f"sched_traj_{kind}_{n}.npz"), init=lat, eps=eps, noise=noise,
                                timesteps=np.asarray(s.timesteps[:3], np.float32), traj=np.stack(traj))


if __name__ == "__main__":
    main()
