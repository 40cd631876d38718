// scheduler.h — host-side scheduler state: timestep/sigma tables and per-step coefficients for the fused
// CFG + scheduler-step kernel.  Table arithmetic restates /root/reference/src/schedulers/*.rs in strict f32
// (this translation unit is built with -ffp-contract=off); each function cites the lines it follows.
#pragma once
#include <vector>

#include "kernels.h"

struct StepPlan {
    int kind = STEP_EULER;      // StepKind
    int pred = SDE_PRED_EPSILON;
    float c[12] = {0};
    int clip = 0;
    int dpm_algo = 0;           // sde_dpm_algorithm
    bool needs_noise = false;   // noise tensor is consumed by this step
    int nhist = 0;              // DPM multistep: buffers in rotation (solver_order) ; LMS: terms in the sum
    float lms_c[4] = {0};
};

struct sde_scheduler {
    int kind = 0;
    sde_sched_cfg cfg{};
    int T = 1000;
    std::vector<float> betas, alphas, alphas_cumprod;
    std::vector<float> sigmas;                      // Euler family (reversed training table + 0, or the interpolated one)
    std::vector<float> alpha_t, sigma_t, lambda_t;  // DPM-Solver
    std::vector<float> timesteps;                   // as f32 (integer-typed schedulers hold exact integers)
    float init_noise_sigma = 1.0f;
    float final_alpha_cumprod = 1.0f;
    int num_inference_steps = -1;
    bool scale_called = false;
    int lower_order_nums = 0;
    int n_model_outputs = 0;   // length of the reference's `model_outputs` deque (only ever grows, dpm_solver_multistep.rs:363-367)
    int lms_hist = 0;          // derivatives stored so far (LMS)
    // device-side history used by the host-mirror step (sde_scheduler_step); txt2img keeps its own
    std::vector<float*> mirror_hist;
    size_t mirror_hist_n = 0;

    void init(int kind, const sde_sched_cfg& cfg);
    void set_timesteps(int n);
    int find(float t) const;                        // `.iter().position(|&p| p == timestep)`
    float input_scale_divisor(float t);             // scale_model_input divides by this; sets scale_called for the Euler family
    bool scales_input() const { return kind == SDE_SCHED_EULER || kind == SDE_SCHED_EULER_A || kind == SDE_SCHED_LMS; }
    bool is_int_timestep() const { return kind == SDE_SCHED_DPMPP || kind == SDE_SCHED_DDIM; }
    void plan(float t, StepPlan& p);                // coefficients for the step at timestep t (no state change)
    void advance(const StepPlan& p);                // state change the reference's `step` makes
    // after a step's kernel: bring the history buffers into the order the next step's kernel expects (the slot the
    // kernel just wrote becomes hist[0])
    void rotate_history(const StepPlan& p, float** hist) const;
    void add_noise_coeffs(float t, float& a, float& b);  // out = a*x0 + b*noise
    ~sde_scheduler();
};
