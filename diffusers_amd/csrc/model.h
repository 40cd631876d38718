// model.h — native UNet2DConditionModel and AutoencoderKL decoder (device weights + forward graphs).
// Layer semantics: diffusers 0.14.0 as exported by /root/reference/scripts/sd2pyke.py.
#pragma once
#include <string>
#include <vector>

#include "kernels.h"
#include "loader.h"

// bump allocator over one device buffer; `dry` mode only measures the peak
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = false;
    void* alloc(size_t bytes)
    {
        size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        if (off > peak) peak = off;
        if (dry) return (void*)(uintptr_t)(a + 4096);  // fake, never dereferenced
        SDE_REQUIRE(off <= cap, SDE_ERR_DEVICE, "activation arena overflow");
        return base + a;
    }
    template <typename T> T* get(size_t n) { return (T*)alloc(n * sizeof(T)); }
    // n ACTIVATION elements: fp16 in the default engine, floats in the fp32 engine (LaunchCtx::f32); the pointer type stays half_t*
    size_t act_bytes = 2;
    half_t* act(size_t n) { return (half_t*)alloc(n * act_bytes); }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

struct DevLinear {
    half_t* W = nullptr;
    float* bias = nullptr;
    int N = 0, Npad = 0, K = 0, taps = 1, Cin = 0;  // Cin = (padded) channels per tap
    bool geglu = false, geglu_pair = false;   // geglu_pair: value/gate rows interleaved in fours (EPI_GEGLU_PAIR)
    bool gelu_tanh = false;                   // GEGLU with FastGelu's tanh form (optimizer-fused model dirs)
};
struct DevNorm { float* gamma = nullptr; float* beta = nullptr; int C = 0; };

struct ResnetW {
    DevNorm n1, n2;
    DevLinear conv1, conv2, shortcut;
    // UNet only: the shortcut weight laid out for split-fp16 operands, K = [x.hi | x.lo] against [W | W] (twice the K of a small
    // 1x1 conv instead of rounding the trunk to fp16 on its way through every channel-changing resnet).  sc_c0 = channels of the
    // first concat source (the hidden state); a concat input takes two chained GEMMs, one per source (shortcut_a2 then shortcut_b2).
    DevLinear shortcut_a2, shortcut_b2;
    int sc_c0 = 0;
    bool has_sc2 = false;
    bool has_shortcut = false, has_temb = false;
    int cin = 0, cout = 0, temb_off = 0;
};
struct TransformerW {
    DevNorm gn, ln1, ln2, ln3;
    DevLinear proj_in, proj_out, qkv, out1, q2, kv2, out2, ff1, ff2;
    DevLinear proj_out2;   // proj_out for a split-fp16 operand: K = [h.hi | h.lo] against [W | W]
    int C = 0, heads = 0, kv_index = 0;
};
struct VaeAttnW {
    DevNorm gn;
    DevLinear q, k, v, proj, qkv;
    int C = 0;
};

struct Exec {
    LaunchCtx ctx;
    Arena* arena = nullptr;
    // UNet trunk (the tensor that runs resnet -> transformer -> resnet ..., the skip connections and the shortcut outputs) kept as
    // split fp16 (hi + lo, kernels.h) wherever a level has at most split_max_hw pixels; 0 = plain fp16 everywhere (VAE decoder).
    long long split_max_hw = 0;
    bool dry() const { return arena->dry; }
    bool split(int HW) const { return split_max_hw > 0 && HW <= split_max_hw; }
};

struct WeightBuilder;  // model.cpp

struct UNetModel {
    // architecture
    int in_channels = 4, out_channels = 4, n_levels = 0, layers_per_block = 0, ctx_dim = 0, temb_dim = 0;
    bool linear_proj = false;
    std::vector<int> boc, heads;
    std::vector<bool> down_attn;
    int64_t n_params = 0;
    // weights
    DevLinear conv_in, conv_out, time_l1, time_l2, temb_all;
    DevNorm norm_out;
    struct DownBlock { std::vector<ResnetW> res; std::vector<TransformerW> att; DevLinear down; bool has_down = false; };
    struct UpBlock { std::vector<ResnetW> res; std::vector<TransformerW> att; DevLinear up; bool has_up = false; };
    std::vector<DownBlock> down;
    std::vector<UpBlock> up;
    ResnetW mid_res0, mid_res1;
    TransformerW mid_att;
    int temb_total = 0, n_transformers = 0;
    std::vector<TransformerW*> transformers;  // in kv_index order

    void build(const OnnxModel& m, const WeightMap& w, WeightBuilder& b, int heads_override);
    // kv_cache[i]: [N*S][2*C_i] per transformer; if fill_kv, (re)computed from ehs_h here.  t_dev: device float, the truncated timestep
    void forward(Exec& ex, const half_t* x_packed, int N, int h, int w, const float* t_dev, const half_t* ehs_h, int S,
                 half_t* const* kv_cache, bool fill_kv, float* out_nchw) const;
    size_t kv_elems(int i, int N, int S) const { return (size_t)N * S * 2 * transformers[i]->C; }
    static constexpr int kInPad = 64;  // sample channels padded to one 64-wide K chunk
};

struct VaeDecoderModel {
    int latent_channels = 4, out_channels = 3, n_levels = 0, layers_per_block = 0;
    std::vector<int> boc;  // decoder order (widest first)
    int64_t n_params = 0;
    DevLinear post_quant, conv_in, conv_out;
    DevNorm norm_out;
    ResnetW mid_res0, mid_res1;
    VaeAttnW mid_att;
    struct UpBlock { std::vector<ResnetW> res; DevLinear up; bool has_up = false; };
    std::vector<UpBlock> up;

    void build(const OnnxModel& m, const WeightMap& w, WeightBuilder& b);
    // x_packed: [B*h*w][64] fp16 latents (already scaled); epi = EPI_IMAGE or EPI_NCHW_F32
    void forward(Exec& ex, const half_t* x_packed, int B, int h, int w, int epi, float* out) const;
    int upscale() const { return 1 << (n_levels - 1); }
    static constexpr int kInPad = 64;
};

// AutoencoderKL encoder + quant_conv (vae_encoder.onnx, sd2pyke.py:181-187,325-346): the session the reference opens
// (impl_main.rs:58,140-150) and never runs; here the start of an img2img run.
struct VaeEncoderModel {
    int in_channels = 3, latent_channels = 4, n_levels = 0, layers_per_block = 0;
    std::vector<int> boc;
    int64_t n_params = 0;
    DevLinear conv_in, conv_out, quant;
    DevNorm norm_out;
    struct DownBlock { std::vector<ResnetW> res; DevLinear down; bool has_down = false; };
    std::vector<DownBlock> down;
    ResnetW mid_res0, mid_res1;
    VaeAttnW mid_att;

    void build(const OnnxModel& m, const WeightMap& w, WeightBuilder& b);
    // x_packed: [B*H*W][64] images in [-1, 1] (3 channels + zero pad); moments: f32 NCHW [B, 2*latent, h, w] (mean | logvar)
    void forward(Exec& ex, const half_t* x_packed, int B, int H, int W, float* moments) const;
    int downscale() const { return 1 << (n_levels - 1); }
    static int out_size(int X, int levels) { for (int i = 0; i + 1 < levels; ++i) X = (X - 2) / 2 + 1; return X; }
    static constexpr int kInPad = 64;
};

// uploads + repacks weights; owns every device allocation it makes
struct WeightBuilder {
    bool gelu_tanh = false;     // the model being built came from the optimizer with --approximate-gelu (FastGelu nodes)
    LaunchCtx ctx;
    const WeightMap* w = nullptr;
    std::vector<void*> allocs;
    void* staging = nullptr;
    size_t staging_cap = 0;
    int64_t params = 0;

    ~WeightBuilder();
    void* dmalloc(size_t bytes);
    const BoundTensor& get(const std::string& name) const;
    bool has(const std::string& name) const { return w->count(name) != 0; }
    // rows of `names` are concatenated along N; cin_pad pads the per-tap channel count
    DevLinear linear(const std::vector<std::string>& names, const std::vector<std::string>& bias_names, int cin_pad = 0,
                     bool geglu = false);
    // 1x1 / Linear weight for a split-fp16 operand: input channels [i0, i0 + n) of `name` laid out twice along K ([W | W], K = 2n)
    DevLinear linear_dup(const std::string& name, const std::vector<std::string>& bias_names, int i0, int n);
    DevNorm norm(const std::string& prefix);
    std::vector<float> host_f32(const std::string& name) const;
    float* upload_f32(const std::vector<float>& v);
    void release_staging();
    void disown() { allocs.clear(); }
};
