#!/usr/bin/env python3
"""Scheduler sweep on the GPU: every scheduler kind x step counts x beta schedules (x prediction types where the reference has
them), full trajectories of the fused step kernel (through the C ABI's scheduler mirror) against the oracle's strict-f32
restatement.  Bar as in tests/test_gpu_schedulers.py.  usage: python tools/fuzz_schedulers.py"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
from diffusers_amd import schedulers as N
from oracle import schedulers as S


def main():
    rng = np.random.default_rng(0)
    bad = cnt = 0
    for kind in ["euler", "euler_a", "dpmpp", "ddim", "ddpm", "lms"]:
        preds = ["epsilon"] + (["v_prediction", "sample"] if kind in ("dpmpp", "ddim") else (["sample"] if kind == "ddpm" else []))
        for n in [1, 2, 3, 7, 20, 50, 100, 333, 1000]:
            for sched in ["scaled_linear", "linear"]:
                for pred in preds:
                    kw_o = dict(T=1000, schedule=sched)
                    kw_n = dict(num_train_timesteps=1000, beta_schedule=sched)
                    if pred != "epsilon":
                        kw_o["prediction_type"] = pred; kw_n["prediction_type"] = pred
                    try:
                        o = S.make(kind, **kw_o); o.set_timesteps(n)
                    except Exception:
                        continue
                    s = N.BY_NAME[kind].stable_diffusion_v1_optimized_default(**kw_n); s.set_timesteps(n)
                    x = xo = (rng.standard_normal((1, 4, 8, 8)).astype(np.float32) * o.init_noise_sigma()).astype(np.float32)
                    ts = list(o.timesteps)
                    if len(ts) > 40:      # long schedules: first 20 and last 20 steps (the state carries over in between for both sides)
                        idx = set(range(20)) | set(range(len(ts) - 20, len(ts)))
                    else:
                        idx = set(range(len(ts)))
                    ok = True
                    for i, t in enumerate(ts):
                        e = rng.standard_normal(x.shape).astype(np.float32)
                        nz = rng.standard_normal(x.shape).astype(np.float32)
                        s.scale_model_input(x, float(t)); o.scale_model_input(xo, t)
                        x, xo = s.step(e, float(t), x, nz), o.step(e, t, xo, nz)
                        if i in idx and not (np.abs(x - xo).max() <= 1e-5 * max(1.0, np.abs(xo).max())):
                            ok = False
                            print("MISS", kind, n, sched, pred, "step", i, float(np.abs(x - xo).max()), float(np.abs(xo).max()), flush=True)
                            break
                        if not np.isfinite(xo).all():
                            break
                    cnt += 1
                    bad += 0 if ok else 1
    print(f"{cnt} trajectories, {bad} misses")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
