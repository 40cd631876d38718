"""CPU: the oracle's scheduler restatement against closed-form identities and SURVEY.md §3.3 sanity values,
and the native library's host-side tables against the oracle (bit-exact: both are strict f32)."""
import numpy as np
import pytest

from oracle import schedulers as S


def test_sd_v1_tables():
    e = S.make("euler")
    assert abs(e.alphas_cumprod[0] - 0.99915) < 1e-6
    assert abs(e.alphas_cumprod[999] - 0.0046601) < 1e-7
    assert abs(e.init_noise_sigma() - 14.61466) < 1e-4
    e.set_timesteps(20)
    assert np.allclose(e.timesteps[:4], [999, 946.421, 893.842, 841.263], atol=1e-3)
    assert np.allclose(e.sigmas[:4], [14.6147, 10.7469, 8.0815, 6.2050], atol=1e-3)
    assert np.allclose(e.sigmas[-3:], [0.23217, 0.02917, 0.0], atol=1e-4)


def test_reference_quirks_timesteps():
    d = S.make("ddim"); d.set_timesteps(50)
    assert len(d.timesteps) == 49 and d.timesteps[0] == 961 and d.timesteps[-1] == 1       # ddim.rs:184
    d.set_timesteps(30)
    assert len(d.timesteps) == 29 and d.timesteps[0] == 925
    p = S.make("ddpm"); p.set_timesteps(30)
    assert len(p.timesteps) == 31 and p.timesteps[0] == 990 and p.timesteps[-1] == 0       # ddpm.rs:181
    m = S.make("dpmpp"); m.set_timesteps(30)
    assert list(m.timesteps[:5]) == [999, 964, 930, 895, 861] and len(m.timesteps) == 30  # `as usize` truncation
    assert np.all(S.betas_for_alpha_bar(10, 0.999)[:-1] == 0)                              # integer division quirk


def test_closed_form_identities():
    rng = np.random.default_rng(0)
    s = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    e = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    # Euler: prev = s + (sigma' - sigma) * eps
    sch = S.make("euler"); sch.set_timesteps(20)
    t = sch.timesteps[3]; sch.scale_model_input(s, t)
    out = sch.step(e, t, s)
    assert np.allclose(out, s + (sch.sigmas[4] - sch.sigmas[3]) * e, rtol=1e-5, atol=1e-5)
    # Euler-a last step (sigma_to = 0): prev = x0
    sa = S.make("euler_a"); sa.set_timesteps(20)
    t = sa.timesteps[-1]; sa.scale_model_input(s, t)
    out = sa.step(e, t, s, noise=rng.standard_normal(s.shape).astype(np.float32))
    assert np.allclose(out, s - sa.sigmas[19] * e, rtol=1e-5, atol=1e-6)
    # DPM++ last step: t = 0 -> prev_t = 0 -> h = 0 -> prev = s
    dp = S.make("dpmpp"); dp.set_timesteps(30)
    assert np.allclose(dp.step(e, 0, s), s, atol=1e-6)
    # DPM++ first step coefficients
    cs, cx, h = dp.first_order_coeffs(999, 964)
    assert abs(cs - 0.998806) < 1e-5 and abs(-cx - 0.015698) < 1e-5 and abs(h - 0.20720) < 1e-4
    # DDIM eta=0 is deterministic and reduces to sqrt(a_prev)*x0 + sqrt(1-a_prev)*eps
    dd = S.make("ddim"); dd.set_timesteps(50)
    a_t, a_p = dd.coeffs(961)
    x0 = (s - np.sqrt(1 - a_t) * e) / np.sqrt(a_t)
    assert np.allclose(dd.step(e, 961, s), np.sqrt(a_p) * x0 + np.sqrt(1 - a_p) * e, rtol=1e-5, atol=1e-5)
    # LMS coefficients integrate the Lagrange basis: they sum to sigma_{i+1} - sigma_i
    lm = S.make("lms"); lm.set_timesteps(25)
    for i in (0, 1, 5, 24):
        assert abs(sum(map(float, lm.lms_coeffs(i))) - float(lm.sigmas[i + 1] - lm.sigmas[i])) < 1e-4


def test_ctor_errors():
    with pytest.raises(ValueError):
        S.EulerDiscrete(T=0)
    with pytest.raises(ValueError):
        S.EulerDiscrete(beta_start=0.02, beta_end=0.01)
    with pytest.raises(ValueError):
        S.EulerDiscrete(beta_start=0.0)


@pytest.mark.parametrize("kind", ["euler", "euler_a", "dpmpp", "ddim", "ddpm", "lms"])
@pytest.mark.parametrize("n", [1, 2, 20, 25, 30, 50])
def test_native_tables_bit_exact(kind, n):
    """Host-side (no GPU) part of the C ABI: timesteps / sigma tables / init sigma == oracle, bit for bit."""
    from diffusers_amd import schedulers as N
    o = S.make(kind); o.set_timesteps(n)
    s = N.BY_NAME[kind].stable_diffusion_v1_optimized_default(); s.set_timesteps(n)
    assert np.array_equal(s.timesteps(), np.asarray(o.timesteps, np.float32))
    assert np.float32(s.init_noise_sigma()) == np.float32(o.init_noise_sigma())
    if kind in ("euler", "euler_a", "lms"):
        assert np.array_equal(s.sigmas(), o.sigmas)


def test_native_scheduler_errors():
    from diffusers_amd import _lib, schedulers as N
    with pytest.raises(_lib.SdeError):
        N.EulerDiscreteScheduler(num_train_timesteps=0)
    with pytest.raises(_lib.SdeError):
        N.DDIMScheduler(beta_start=0.5, beta_end=0.1)
    d = N.EulerDiscreteScheduler()                     # Default: linear 0.0001..0.02 (euler_discrete.rs:41-45)
    o = S.EulerDiscrete(1000, 0.0001, 0.02, "linear")
    assert np.float32(d.init_noise_sigma()) == o.init_noise_sigma()
    assert len(d) == 1000 and d.order() == 1


def test_dpm_solver_and_dpm_solver_pp_first_order_agree():
    """De-correlation check on the two restated first-order formulas (dpm_solver_multistep.rs:254 vs :255): for an epsilon
    model both are the exact exponential-integrator (DDIM) step, so they must agree up to f32 rounding although they share no
    coefficient: (sigma_t/sigma_s) x - alpha_t (e^-h - 1) x0  ==  (alpha_t/alpha_s) x - sigma_t (e^h - 1) eps."""
    rng = np.random.default_rng(0)
    a, b = S.make("dpmpp", algorithm_type="dpmsolver++"), S.make("dpmpp", algorithm_type="dpmsolver")
    a.set_timesteps(30); b.set_timesteps(30)
    xa = xb = rng.standard_normal((1, 4, 8, 8)).astype(np.float32)
    for t in a.timesteps[:-1]:
        e = rng.standard_normal(xa.shape).astype(np.float32)
        xa, xb = a.step(e, t, xa), b.step(e, t, xb)
        assert np.abs(xa - xb).max() <= 2e-4 * max(1.0, np.abs(xa).max())


def test_dpm_multistep_orders_converge_on_a_linear_ode():
    """With a model whose x0 prediction is a constant c the exact solution is known for every order: every update is
    prev = (sigma_t/sigma_s) x + alpha_t (1 - e^-h) c (the d1 / d2 terms vanish), so all orders / solver types agree."""
    rng = np.random.default_rng(1)
    c = rng.standard_normal((1, 4, 4, 4)).astype(np.float32)
    outs = []
    for order, solver in [(1, "midpoint"), (2, "midpoint"), (2, "heun"), (3, "midpoint")]:
        s = S.make("dpmpp", reference_quirks=False, solver_order=order, solver_type=solver, prediction_type="sample")
        s.set_timesteps(20)
        x = np.ones_like(c)
        for t in s.timesteps:
            x = s.step(c, t, x)
        outs.append(x)
    for o in outs[1:]:
        assert np.abs(o - outs[0]).max() < 1e-5
