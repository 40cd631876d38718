// scheduler.cpp — see scheduler.h.  Strict f32, no contraction: every intermediate is rounded as the
// reference's Rust f32 expressions are.
#include "scheduler.h"

#include <algorithm>
#include <cmath>

namespace {
// ndarray 0.15 Array1::linspace(a, b, n): a + i * ((b - a) / (n - 1))
std::vector<float> nd_linspace(float a, float b, int n)
{
    std::vector<float> v((size_t)std::max(n, 0));
    float step = n > 1 ? (b - a) / (float)(n - 1) : 0.0f;
    for (int i = 0; i < n; ++i) { float si = step * (float)i; v[(size_t)i] = a + si; }
    return v;
}
// ndarray 0.15 Array1::range(start, end, step): ceil((end - start) / step) items
std::vector<float> nd_range(float start, float end, float step)
{
    float q = (end - start) / step;
    int n = (int)std::ceil(q);
    if (n < 0) n = 0;
    std::vector<float> v((size_t)n);
    for (int i = 0; i < n; ++i) { float si = step * (float)i; v[(size_t)i] = start + si; }
    return v;
}
// schedulers/mod.rs:70-79 — note the INTEGER divisions i / T (REFERENCE_QUIRK)
std::vector<float> betas_for_alpha_bar(int T, float max_beta)
{
    auto alpha_bar = [](int ts) {
        float x = ((float)ts + 0.008f) / 1.008f;
        x = x * 1.57079632679489661923f;
        float c = std::cos(x);
        return c * c;
    };
    std::vector<float> b;
    for (int i = 0; i < T; ++i) {
        int t1 = i / T, t2 = (i + 1) / T;
        float r = alpha_bar(t2) / alpha_bar(t1);
        float v = 1.0f - r;
        b.push_back(std::min(v, max_beta));
    }
    return b;
}
// util/interpolation.rs:42-105
struct Interp {
    const std::vector<float>& x; const std::vector<float>& y; size_t cache = 0;
    static size_t bsearch(const std::vector<float>& xa, float x, size_t lo, size_t hi)
    {
        while (hi > lo + 1) { size_t i = (hi + lo) / 2; if (xa[i] > x) hi = i; else lo = i; }
        return lo;
    }
    float eval(float v)
    {
        if (x.size() < 2 || v < x.front() || v > x.back()) return NAN;
        if (v < x[cache]) cache = bsearch(x, v, 0, cache);
        else if (v >= x[cache + 1]) cache = bsearch(x, v, cache, x.size() - 1);
        size_t i = cache;
        float xl = x[i], xh = x[i + 1], yl = y[i], yh = y[i + 1];
        float dx = xh - xl;
        if (dx > 0.0f) { float a = v - xl; float q = a / dx; float dy = yh - yl; float m = q * dy; return yl + m; }
        return NAN;
    }
};
bool is_normal(float v) { return std::isnormal(v); }
}  // namespace

sde_scheduler::~sde_scheduler()
{
    for (float* p : mirror_hist) if (p) (void)hipFree(p);
}

void sde_scheduler::init(int k, const sde_sched_cfg& c)
{
    kind = k; cfg = c; T = c.num_train_timesteps;
    SDE_REQUIRE(k >= SDE_SCHED_EULER && k <= SDE_SCHED_LMS, SDE_ERR_INVALID_ARGUMENT, "unknown scheduler kind");
    // euler_discrete.rs:64-72 (same checks in all five)
    SDE_REQUIRE(T > 0, SDE_ERR_INVALID_ARGUMENT, "num_train_timesteps (" + std::to_string(T) + ") must be >0");
    SDE_REQUIRE(is_normal(c.beta_start) && is_normal(c.beta_end), SDE_ERR_INVALID_ARGUMENT,
                "beta_start and beta_end must be normal (not zero, infinite, NaN, or subnormal)");
    SDE_REQUIRE(c.beta_start < c.beta_end, SDE_ERR_INVALID_ARGUMENT, "beta_start must be < beta_end");
    if (k == SDE_SCHED_DPMPP) {
        SDE_REQUIRE(c.solver_order >= 1 && c.solver_order <= 3, SDE_ERR_INVALID_ARGUMENT, "solver_order must be 1, 2 or 3");
        SDE_REQUIRE(c.algorithm_type == SDE_DPM_SOLVER_PP || c.algorithm_type == SDE_DPM_SOLVER, SDE_ERR_INVALID_ARGUMENT, "unknown algorithm_type");
        SDE_REQUIRE(c.solver_type == SDE_DPM_MIDPOINT || c.solver_type == SDE_DPM_HEUN, SDE_ERR_INVALID_ARGUMENT, "unknown solver_type");
    }
    // betas (euler_discrete.rs:74-84)
    if (c.beta_schedule == SDE_BETA_LINEAR) betas = nd_linspace(c.beta_start, c.beta_end, T);
    else if (c.beta_schedule == SDE_BETA_SCALED_LINEAR) {
        betas = nd_linspace(std::sqrt(c.beta_start), std::sqrt(c.beta_end), T);
        for (auto& b : betas) b = b * b;
    } else if (c.beta_schedule == SDE_BETA_SQUAREDCOS_CAP_V2) betas = betas_for_alpha_bar(T, 0.999f);
    else sde_fail(SDE_ERR_INVALID_ARGUMENT, "beta schedule not implemented");
    alphas.resize((size_t)T); alphas_cumprod.resize((size_t)T);
    float prod = 1.0f;
    for (int i = 0; i < T; ++i) { alphas[(size_t)i] = 1.0f - betas[(size_t)i]; prod = prod * alphas[(size_t)i]; alphas_cumprod[(size_t)i] = prod; }
    init_noise_sigma = 1.0f;
    num_inference_steps = -1; scale_called = false; lower_order_nums = 0; n_model_outputs = 0; lms_hist = 0;
    if (scales_input()) {
        // euler_discrete.rs:97-109: sigmas reversed + 0 ; timesteps = linspace(T-1, 0, T) ; init sigma = max (never refreshed)
        sigmas.resize((size_t)T + 1);
        for (int i = 0; i < T; ++i) { float ac = alphas_cumprod[(size_t)(T - 1 - i)]; float r = (1.0f - ac) / ac; sigmas[(size_t)i] = std::sqrt(r); }
        sigmas[(size_t)T] = 0.0f;
        timesteps = nd_linspace((float)T - 1.0f, 0.0f, T);
        float mx = sigmas[0];
        for (float s : sigmas) if (s > mx) mx = s;
        init_noise_sigma = mx;
    } else if (kind == SDE_SCHED_DPMPP) {
        // dpm_solver_multistep.rs:186-190
        alpha_t.resize((size_t)T); sigma_t.resize((size_t)T); lambda_t.resize((size_t)T);
        for (int i = 0; i < T; ++i) {
            float ac = alphas_cumprod[(size_t)i];
            alpha_t[(size_t)i] = std::sqrt(ac);
            float om = 1.0f - ac;
            sigma_t[(size_t)i] = std::sqrt(om);
            float la = std::log(alpha_t[(size_t)i]), ls = std::log(sigma_t[(size_t)i]);
            lambda_t[(size_t)i] = la - ls;
        }
        timesteps = nd_linspace((float)T - 1.0f, 0.0f, T);
        for (auto& t : timesteps) t = (float)(long long)t;  // `as usize`
    } else if (kind == SDE_SCHED_DDIM) {
        // ddim.rs:132-137
        final_alpha_cumprod = c.set_alpha_to_one ? 1.0f : alphas_cumprod[0];
        auto ts = nd_linspace(0.0f, (float)T - 1.0f, T);
        timesteps.assign(ts.rbegin(), ts.rend());
        for (auto& t : timesteps) t = (float)(long long)t;
    } else {  // DDPM, ddpm.rs:130-132
        auto ts = nd_linspace(0.0f, (float)T - 1.0f, T);
        timesteps.assign(ts.rbegin(), ts.rend());
    }
}

void sde_scheduler::set_timesteps(int n)
{
    SDE_REQUIRE(n >= 1, SDE_ERR_INVALID_ARGUMENT, "num_inference_steps must be >= 1");
    num_inference_steps = n;
    if (scales_input()) {
        // euler_discrete.rs:152-173.  The interpolator indexes x[i + 1]: with a one-entry training table the reference panics on the
        // bounds check (util/interpolation.rs:98-99); here that is an error status
        SDE_REQUIRE(T >= 2, SDE_ERR_SCHEDULER, "sigma interpolation needs num_train_timesteps >= 2");
        timesteps = nd_linspace((float)T - 1.0f, 0.0f, n);
        std::vector<float> sg((size_t)T);
        for (int i = 0; i < T; ++i) { float ac = alphas_cumprod[(size_t)i]; float r = (1.0f - ac) / ac; sg[(size_t)i] = std::sqrt(r); }
        std::vector<float> xa = nd_range(0.0f, (float)T, 1.0f);
        Interp ip{xa, sg};
        sigmas.assign((size_t)n + 1, 0.0f);
        for (int i = 0; i < n; ++i) sigmas[(size_t)i] = ip.eval(timesteps[(size_t)i]);
        sigmas[(size_t)n] = 0.0f;
        lms_hist = 0;
    } else if (kind == SDE_SCHED_DPMPP) {
        // dpm_solver_multistep.rs:340-348
        timesteps = nd_linspace((float)T - 1.0f, 0.0f, n);
        for (auto& t : timesteps) t = (float)(long long)t;
        n_model_outputs = 0; lower_order_nums = 0;
    } else if (kind == SDE_SCHED_DDIM) {
        // ddim.rs:179-190 — REFERENCE_QUIRK: range(0, n-1) is end-exclusive => n-1 timesteps
        SDE_REQUIRE(n <= T, SDE_ERR_INVALID_ARGUMENT, "num_inference_steps exceeds num_train_timesteps");
        int ratio = T / n;
        auto r = nd_range(0.0f, (float)(cfg.reference_quirks ? n - 1 : n), 1.0f);
        timesteps.clear();
        for (auto it = r.rbegin(); it != r.rend(); ++it) {
            float v = std::round(*it * (float)ratio);
            long long iv = (long long)v + cfg.steps_offset;
            timesteps.push_back((float)iv);
        }
    } else {
        // ddpm.rs:177-186
        int nn = std::min(n, T);
        num_inference_steps = nn;
        auto r = nd_range(0.0f, (float)T, (float)(T / nn));
        timesteps.assign(r.rbegin(), r.rend());
    }
}

int sde_scheduler::find(float t) const
{
    for (size_t i = 0; i < timesteps.size(); ++i) if (timesteps[i] == t) return (int)i;
    return -1;
}

float sde_scheduler::input_scale_divisor(float t)
{
    if (!scales_input()) return 1.0f;  // dpm_solver_multistep.rs:336-338, ddim.rs:175-177, ddpm.rs:173-175
    // euler_discrete.rs:134-150
    int i = find(t);
    SDE_REQUIRE(i >= 0, SDE_ERR_SCHEDULER, "timestep out of this schedulers bounds: " + std::to_string(t));
    float s = sigmas[(size_t)i];
    scale_called = true;
    float s2 = s * s;
    return std::sqrt(s2 + 1.0f);
}

void sde_scheduler::plan(float t, StepPlan& p)
{
    p = StepPlan();
    p.pred = cfg.prediction_type;
    const int n = (int)timesteps.size();
    if (scales_input()) {
        SDE_REQUIRE(scale_called, SDE_ERR_SCHEDULER, "scale_model_input must be called before step");  // euler_discrete.rs:181
        int i = find(t);
        SDE_REQUIRE(i >= 0, SDE_ERR_SCHEDULER, "timestep out of this schedulers bounds: " + std::to_string(t));
        float sigma = sigmas[(size_t)i], next = sigmas[(size_t)i + 1];
        p.pred = SDE_PRED_EPSILON;
        if (kind == SDE_SCHED_EULER) {          // euler_discrete.rs:176-214, s_churn = 0 => gamma = 0
            float sigma_hat = sigma * (0.0f + 1.0f);
            p.kind = STEP_EULER; p.c[0] = sigma_hat; p.c[1] = next - sigma_hat;
        } else if (kind == SDE_SCHED_EULER_A) { // euler_ancestral_discrete.rs:184-204
            float sf2 = sigma * sigma, st2 = next * next;
            float num = st2 * (sf2 - st2);
            float up = std::sqrt(num / sf2);
            float dn = std::sqrt(st2 - up * up);
            p.kind = STEP_EULER_A; p.c[0] = sigma; p.c[1] = dn - sigma; p.c[2] = up; p.needs_noise = true;
        } else {                                // LMS (k-diffusion; not in the reference)
            p.kind = STEP_LMS; p.c[0] = sigma;
            int order = std::min(i + 1, 4);
            p.nhist = order;
            for (int k = 0; k < order; ++k) {
                // integral over [sigma_i, sigma_{i+1}] of the Lagrange basis polynomial l_k(tau) on nodes sigma_{i-j}
                double poly[5] = {1, 0, 0, 0, 0}; int deg = 0;  // ascending coefficients
                for (int j = 0; j < order; ++j) {
                    if (j == k) continue;
                    double sj = sigmas[(size_t)(i - j)], denom = (double)sigmas[(size_t)(i - k)] - sj;
                    double nxt[5] = {0, 0, 0, 0, 0};
                    for (int d = 0; d <= deg; ++d) { nxt[d + 1] += poly[d] / denom; nxt[d] += -sj * poly[d] / denom; }
                    ++deg;
                    for (int d = 0; d <= deg; ++d) poly[d] = nxt[d];
                }
                auto P = [&](double x) { double s = 0, xp = x; for (int d = 0; d <= deg; ++d) { s += poly[d] * xp / (d + 1); xp *= x; } return s; };
                p.lms_c[k] = (float)(P((double)next) - P((double)sigma));
            }
        }
        return;
    }
    if (kind == SDE_SCHED_DPMPP) {
        // dpm_solver_multistep.rs:350-380
        int i = find(t);
        SDE_REQUIRE(i >= 0, SDE_ERR_SCHEDULER, "timestep out of this schedulers bounds: " + std::to_string(t));
        int ts = (int)t;
        int prev = (i == n - 1) ? 0 : (int)timesteps[(size_t)i + 1];
        const bool lower_final = (i == n - 1) && cfg.lower_order_final && n < 15;
        const bool lower_second = (i == n - 2) && cfg.lower_order_final && n < 15;
        const bool dpmpp = cfg.algorithm_type == SDE_DPM_SOLVER_PP;
        float lam_t = lambda_t[(size_t)prev], lam_s = lambda_t[(size_t)ts];
        float h = lam_t - lam_s;
        // REFERENCE_QUIRK: lower_order_nums is never incremented there => always first order (:369)
        bool first = cfg.solver_order == 1 || lower_order_nums < 1 || lower_final;
        // "fixed" mode only: the last timestep of linspace(T-1, 0, n) is 0 and its prev_timestep is 0 as well, so h == 0 and the
        // multistep formulas divide by it (r0 = h0 / 0; Heun: 0 / 0).  The first-order update is the identity there.
        if (!cfg.reference_quirks && h == 0.0f) first = true;
        p.dpm_algo = cfg.algorithm_type;
        p.c[0] = alpha_t[(size_t)ts]; p.c[1] = sigma_t[(size_t)ts];
        // first-order coefficients (:248-257); the multistep updates reuse them for the sample and d0 terms
        float ex;   // D++: exp(-h) - 1 ; DPMSolver: exp(h) - 1
        if (dpmpp) {
            ex = std::exp(-h) - 1.0f;
            p.c[2] = sigma_t[(size_t)prev] / sigma_t[(size_t)ts];
            p.c[3] = alpha_t[(size_t)prev] * ex;
        } else {
            ex = std::exp(h) - 1.0f;
            p.c[2] = alpha_t[(size_t)prev] / alpha_t[(size_t)ts];
            p.c[3] = sigma_t[(size_t)prev] * ex;
        }
        const float outer = dpmpp ? alpha_t[(size_t)prev] : sigma_t[(size_t)prev];
        p.nhist = cfg.reference_quirks ? 0 : std::max(2, std::min(cfg.solver_order, 3));
        if (first) p.kind = STEP_DPM1;
        else if (cfg.solver_order == 2 || lower_order_nums < 2 || lower_second) {  // :371-373, :259-285
            int s1 = (int)timesteps[(size_t)i - 1];
            float h0 = lam_s - lambda_t[(size_t)s1];
            float r0 = h0 / h;
            p.kind = STEP_DPM2M; p.c[4] = 1.0f / r0;
            if (cfg.solver_type == SDE_DPM_MIDPOINT) p.c[5] = 0.5f * p.c[3];                       // :278,282
            else if (dpmpp) { float q = ex / h + 1.0f; p.c[5] = -(outer * q); }                    // :279 (added there: passed negated)
            else { float q = ex / h - 1.0f; p.c[5] = outer * q; }                                  // :283
        } else {                                                                                   // :374-376, :288-326
            int s1 = (int)timesteps[(size_t)i - 1], s2 = (int)timesteps[(size_t)i - 2];
            float lam_s1 = lambda_t[(size_t)s1], lam_s2 = lambda_t[(size_t)s2];
            float h0 = lam_s - lam_s1, h1 = lam_s1 - lam_s2;
            float r0 = h0 / h, r1 = h1 / h;
            p.kind = STEP_DPM3M;
            p.c[4] = 1.0f / r0; p.c[6] = 1.0f / r1; p.c[7] = r0 / (r0 + r1); p.c[8] = 1.0f / (r0 + r1);
            float h2 = h * h;
            if (dpmpp) {
                float q1 = ex / h + 1.0f; p.c[5] = -(outer * q1);                                  // :316 (added there)
                float q2 = (ex + h) / h2 - 0.5f; p.c[9] = outer * q2;                              // :317
            } else {
                float q1 = ex / h - 1.0f; p.c[5] = outer * q1;                                     // :322
                float q2 = (ex - h) / h2 - 0.5f; p.c[9] = outer * q2;                              // :323
            }
        }
        return;
    }
    if (kind == SDE_SCHED_DDIM) {
        // ddim.rs:192-251, ETA = 0
        SDE_REQUIRE(num_inference_steps > 0, SDE_ERR_SCHEDULER, "set_timesteps must be called before step");  // :197 unwrap
        int ts = (int)t;
        SDE_REQUIRE(ts >= 0 && ts < T && (float)ts == t, SDE_ERR_SCHEDULER, "timestep out of this schedulers bounds: " + std::to_string(t));
        int prev = ts - T / num_inference_steps;
        float a_t = alphas_cumprod[(size_t)ts];
        float a_prev = prev >= 0 ? alphas_cumprod[(size_t)prev] : final_alpha_cumprod;
        float b_t = 1.0f - a_t;
        float std_dev = 0.0f;
        float inner = 1.0f - a_prev - std_dev * std_dev;
        p.kind = STEP_DDIM; p.c[0] = std::sqrt(a_t); p.c[1] = std::sqrt(b_t); p.c[2] = std::sqrt(a_prev); p.c[3] = std::sqrt(inner);
        p.clip = cfg.clip_sample;
        return;
    }
    // DDPM, ddpm.rs:188-239 — REFERENCE_QUIRK: tables are indexed by the POSITION of t in `timesteps`
    int pos = find(t);
    SDE_REQUIRE(pos >= 0, SDE_ERR_SCHEDULER, "timestep out of this schedulers bounds: " + std::to_string(t));
    SDE_REQUIRE(cfg.prediction_type != SDE_PRED_V, SDE_ERR_SCHEDULER, "DDPM v-prediction is unimplemented in the reference (ddpm.rs:206)");
    int idx = cfg.reference_quirks ? pos : (int)t;
    float a_t = alphas_cumprod[(size_t)idx];
    float a_prev = idx > 0 ? alphas_cumprod[(size_t)idx - 1] : 1.0f;
    float b_t = 1.0f - a_t, b_prev = 1.0f - a_prev;
    p.kind = STEP_DDPM; p.clip = cfg.clip_sample;
    p.c[0] = std::sqrt(a_t); p.c[1] = std::sqrt(b_t);
    { float num = std::sqrt(a_prev) * betas[(size_t)idx]; p.c[2] = num / b_t; }
    { float num = std::sqrt(alphas[(size_t)idx]) * b_prev; p.c[3] = num / b_t; }
    if (idx > 0) {
        // get_variance :150-163
        float var = (1.0f - a_prev) / (1.0f - a_t) * betas[(size_t)idx];
        float v;
        switch (cfg.variance_type) {
        case 0: v = std::max(var, 1e-20f); break;
        case 1: v = std::log(std::max(var, 1e-20f)); break;
        case 2: v = betas[(size_t)idx]; break;
        case 3: v = std::log(betas[(size_t)idx]); break;
        default: v = var; break;
        }
        p.c[4] = cfg.variance_type == 1 ? v : std::sqrt(v);
        p.needs_noise = true;
    }
}

void sde_scheduler::advance(const StepPlan& p)
{
    if (kind == SDE_SCHED_DPMPP) {
        ++n_model_outputs;
        if (!cfg.reference_quirks && lower_order_nums < cfg.solver_order) ++lower_order_nums;
    } else if (kind == SDE_SCHED_LMS) {
        lms_hist = std::min(lms_hist + 1, 4);
    }
    (void)p;
}

void sde_scheduler::rotate_history(const StepPlan& p, float** hist) const
{
    // DPM multistep: the kernel wrote the converted output into hist[nhist-1]; it becomes hist[0] (m1 of the next step)
    if (kind == SDE_SCHED_DPMPP && p.nhist > 1) std::rotate(hist, hist + p.nhist - 1, hist + p.nhist);
    // LMS: the kernel wrote the new derivative into hist[0] and read hist[1..3]; shift so that hist[0] is the free slot again
    if (kind == SDE_SCHED_LMS) std::rotate(hist, hist + 3, hist + 4);
}

void sde_scheduler::add_noise_coeffs(float t, float& a, float& b)
{
    if (scales_input()) {  // euler_discrete.rs:223-241
        int i = find(t);
        SDE_REQUIRE(i >= 0, SDE_ERR_SCHEDULER, "timestep out of this schedulers bounds: " + std::to_string(t));
        a = 1.0f; b = sigmas[(size_t)i];
        return;
    }
    int idx;
    if (kind == SDE_SCHED_DDPM) {  // ddpm.rs:241-249 (position indexed)
        idx = find(t);
        SDE_REQUIRE(idx >= 0, SDE_ERR_SCHEDULER, "timestep out of this schedulers bounds: " + std::to_string(t));
    } else {                       // dpm_solver_multistep.rs:382-384, ddim.rs:253-255
        idx = (int)t;
        SDE_REQUIRE(idx >= 0 && idx < T, SDE_ERR_SCHEDULER, "timestep out of this schedulers bounds: " + std::to_string(t));
    }
    float ac = alphas_cumprod[(size_t)idx];
    a = std::sqrt(ac); b = std::sqrt(1.0f - ac);
}
