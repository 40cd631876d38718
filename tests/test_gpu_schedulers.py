"""GPU: the fused step kernel (through the C ABI's scheduler mirror) against the oracle's f32 restatement and the
committed golden trajectories.  Bar: f32 element-wise arithmetic in the reference's operation order => agreement to a few
ulp; asserted as max |diff| <= 4e-6 * max|x| (exp/sqrt on the host side are the only libm-dependent scalars)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KINDS = ["euler", "euler_a", "dpmpp", "ddim", "ddpm", "lms"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n", [20, 30])
def test_step_matches_golden_trajectory(kind, n):
    from diffusers_amd import schedulers as N
    g = np.load(os.path.join(GOLD, f"sched_traj_{kind}_{n}.npz"))
    s = N.BY_NAME[kind].stable_diffusion_v1_optimized_default()
    s.set_timesteps(n)
    x = g["init"]
    for i in range(3):
        t = float(g["timesteps"][i])
        s.scale_model_input(x, t)
        x = s.step(g["eps"][i], t, x, g["noise"][i])
        ref = g["traj"][i]
        assert np.abs(x - ref).max() <= 4e-6 * np.abs(ref).max(), (kind, n, i)


@pytest.mark.parametrize("kind", KINDS)
def test_full_trajectory_vs_oracle(kind):
    from diffusers_amd import schedulers as N
    from oracle import schedulers as S
    rng = np.random.default_rng(3)
    o = S.make(kind); o.set_timesteps(25)
    s = N.BY_NAME[kind].stable_diffusion_v1_optimized_default(); s.set_timesteps(25)
    x = xo = (rng.standard_normal((2, 4, 16, 16)).astype(np.float32) * o.init_noise_sigma()).astype(np.float32)
    for t in o.timesteps:
        e = rng.standard_normal(x.shape).astype(np.float32)
        nz = rng.standard_normal(x.shape).astype(np.float32)
        a, b = s.scale_model_input(x, float(t)), o.scale_model_input(xo, t)
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
        x, xo = s.step(e, float(t), x, nz), o.step(e, t, xo, nz)
        assert np.abs(x - xo).max() <= 1e-5 * max(1.0, np.abs(xo).max()), (kind, t)


@pytest.mark.parametrize("kind,pred", [("dpmpp", "v_prediction"), ("ddim", "v_prediction"), ("dpmpp", "sample"), ("ddim", "sample"),
                                       ("ddpm", "sample")])
def test_prediction_types(kind, pred):
    from diffusers_amd import schedulers as N
    from oracle import schedulers as S
    rng = np.random.default_rng(4)
    o = S.make(kind, prediction_type=pred); o.set_timesteps(10)
    s = N.BY_NAME[kind].stable_diffusion_v1_optimized_default(prediction_type=pred); s.set_timesteps(10)
    x = xo = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    for t in o.timesteps[:4]:
        e = rng.standard_normal(x.shape).astype(np.float32)
        nz = rng.standard_normal(x.shape).astype(np.float32)
        x, xo = s.step(e, float(t), x, nz), o.step(e, t, xo, nz)
        assert np.abs(x - xo).max() <= 1e-5 * max(1.0, np.abs(xo).max())


def test_true_dpmpp_2m_mode():
    """reference_quirks=0: the multistep second-order update the reference never reaches."""
    from diffusers_amd import schedulers as N
    from oracle import schedulers as S
    rng = np.random.default_rng(6)
    o = S.make("dpmpp", reference_quirks=False); o.set_timesteps(20)
    s = N.DPMSolverMultistepScheduler.stable_diffusion_v1_optimized_default(reference_quirks=0); s.set_timesteps(20)
    x = xo = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    for t in o.timesteps:
        e = rng.standard_normal(x.shape).astype(np.float32)
        x, xo = s.step(e, float(t), x), o.step(e, t, xo)
        assert np.abs(x - xo).max() <= 2e-5 * max(1.0, np.abs(xo).max())


@pytest.mark.parametrize("algo", ["dpmsolver++", "dpmsolver"])
@pytest.mark.parametrize("pred", ["epsilon", "sample", "v_prediction"])
def test_dpm_algorithm_type_in_parity_mode(algo, pred):
    """DPMSolverAlgorithmType (dpm_solver_multistep.rs:30-36) is live in the reference even with its first-order quirk:
    convert_model_output (:232-244) and the first-order formula (:255)."""
    from diffusers_amd import schedulers as N
    from oracle import schedulers as S
    rng = np.random.default_rng(16)
    o = S.make("dpmpp", algorithm_type=algo, prediction_type=pred); o.set_timesteps(12)
    s = N.DPMSolverMultistepScheduler.stable_diffusion_v1_optimized_default(algorithm_type=algo, prediction_type=pred); s.set_timesteps(12)
    x = xo = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    for t in o.timesteps:
        e = rng.standard_normal(x.shape).astype(np.float32)
        x, xo = s.step(e, float(t), x), o.step(e, t, xo)
        assert np.abs(x - xo).max() <= 1e-5 * max(1.0, np.abs(xo).max()), (algo, pred, t)


@pytest.mark.parametrize("order,algo,solver,n", [(2, "dpmsolver++", "heun", 20), (2, "dpmsolver", "midpoint", 20), (2, "dpmsolver", "heun", 20),
                                                 (3, "dpmsolver++", "midpoint", 20), (3, "dpmsolver", "midpoint", 20),
                                                 (3, "dpmsolver++", "midpoint", 10), (2, "dpmsolver++", "midpoint", 10)])
def test_dpm_multistep_variants(order, algo, solver, n):
    """reference_quirks=0: the second-order (Midpoint / Heun, :259-285) and third-order (:288-326) multistep updates of both
    algorithm types, which the reference's code describes and never reaches; n < 15 exercises lower_order_final (:359-360)."""
    from diffusers_amd import schedulers as N
    from oracle import schedulers as S
    rng = np.random.default_rng(60 + order)
    o = S.make("dpmpp", reference_quirks=False, solver_order=order, algorithm_type=algo, solver_type=solver); o.set_timesteps(n)
    s = N.DPMSolverMultistepScheduler.stable_diffusion_v1_optimized_default(reference_quirks=0, solver_order=order, algorithm_type=algo,
                                                                            solver_type=solver)
    s.set_timesteps(n)
    x = xo = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    for t in o.timesteps:
        e = (rng.standard_normal(x.shape) * 0.5).astype(np.float32)
        x, xo = s.step(e, float(t), x), o.step(e, t, xo)
        assert np.isfinite(xo).all()
        assert np.abs(x - xo).max() <= 5e-5 * max(1.0, np.abs(xo).max()), (order, algo, solver, t)


def test_scheduler_misuse_errors():
    from diffusers_amd import _lib, schedulers as N
    s = N.EulerDiscreteScheduler.stable_diffusion_v1_optimized_default(); s.set_timesteps(20)
    x = np.zeros((1, 4, 8, 8), np.float32)
    with pytest.raises(_lib.SdeError):   # step before scale_model_input (euler_discrete.rs:181 assert)
        s.step(x, 999.0, x)
    with pytest.raises(_lib.SdeError):   # timestep not in the table (euler_discrete.rs:135-140 unwrap)
        s.scale_model_input(x, 123.456)
    a = N.EulerAncestralDiscreteScheduler.stable_diffusion_v1_optimized_default(); a.set_timesteps(20)
    a.scale_model_input(x, 999.0)
    with pytest.raises(_lib.SdeError):   # ancestral step without noise
        a.step(x, 999.0, x)
    d = N.DDPMScheduler.stable_diffusion_v1_optimized_default(prediction_type="v_prediction"); d.set_timesteps(20)
    with pytest.raises(_lib.SdeError):   # ddpm.rs:206 unimplemented!()
        d.step(x, float(d.timesteps()[0]), x, x)
    with pytest.raises(_lib.SdeError, match="solver_order"):
        N.DPMSolverMultistepScheduler.stable_diffusion_v1_optimized_default(solver_order=4)
    with pytest.raises(_lib.SdeError):   # one training timestep: the reference's interpolator panics on its bounds check
        N.EulerDiscreteScheduler(num_train_timesteps=1).set_timesteps(1)


def test_add_noise():
    from diffusers_amd import schedulers as N
    from oracle import schedulers as S
    rng = np.random.default_rng(8)
    x, nz = rng.standard_normal((2, 1, 4, 8, 8)).astype(np.float32)
    for kind in KINDS:
        o = S.make(kind); o.set_timesteps(20)
        s = N.BY_NAME[kind].stable_diffusion_v1_optimized_default(); s.set_timesteps(20)
        t = o.timesteps[5]
        assert np.abs(s.add_noise(x, nz, float(t)) - o.add_noise(x, nz, t)).max() < 1e-5


@pytest.mark.parametrize("kind,pred", [("euler", None), ("euler_a", None), ("lms", None), ("ddim", "epsilon"), ("ddim", "v_prediction"),
                                       ("ddpm", "epsilon"), ("dpmpp", "epsilon")])
def test_step_output_optional_fields(kind, pred):
    """SchedulerStepOutput (schedulers/mod.rs:109-150): `pred_original_sample` where the reference's step returns Some(..)
    (Euler / Euler-a / DDIM / DDPM), None for DPM-Solver (dpm_solver_multistep.rs:379); the Euler family's derivative on request."""
    from diffusers_amd import schedulers as N
    from oracle import schedulers as S
    rng = np.random.default_rng(21)
    kw = {} if pred is None else dict(prediction_type=pred)
    o = S.make(kind, **kw); o.set_timesteps(12)
    s = N.BY_NAME[kind].stable_diffusion_v1_optimized_default(**kw); s.set_timesteps(12)
    x = xo = (rng.standard_normal((1, 4, 8, 8)).astype(np.float32) * o.init_noise_sigma()).astype(np.float32)
    for t in o.timesteps[:5]:
        e = rng.standard_normal(x.shape).astype(np.float32)
        nz = rng.standard_normal(x.shape).astype(np.float32)
        s.scale_model_input(x, float(t)); o.scale_model_input(xo, t)
        out = s.step_output(e, float(t), x, nz)
        xo = o.step(e, t, xo, nz)
        x = out.prev_sample()
        assert np.abs(x - xo).max() <= 1e-5 * max(1.0, np.abs(xo).max())
        assert out.prev_sample_mean() is None
        want = o.last_step
        if want["pred_original_sample"] is None:
            assert out.pred_original_sample() is None and kind == "dpmpp"
        else:
            assert np.abs(out.pred_original_sample() - want["pred_original_sample"]).max() <= 1e-5 * max(1.0, np.abs(want["pred_original_sample"]).max())
        if want["derivative"] is None:
            assert out.derivative() is None
        else:
            assert np.abs(out.derivative() - want["derivative"]).max() <= 1e-5 * max(1.0, np.abs(want["derivative"]).max())
