"""Generates the committed golden fixtures.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

* vae_decoder.onnx      — copy of the reference's own test fixture data file
                          (/root/reference/tests/stable-diffusion/vae_decoder.onnx, a tiny random VAE exported by
                          torch 1.13.1; it is a data file the reference's tests hold, not source).
* vae_fixture_io.npz    — seeded latent + the output of that graph executed by oracle/onnx_interp.py
                          (ONNX Runtime is not available here, so the graph is interpreted op by op).
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
