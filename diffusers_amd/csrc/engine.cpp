// engine.cpp — the C ABI (include/sdengine.h): engine lifecycle, the two "session.run" replacements,
// the on-device txt2img loop, scheduler handles, measurement hooks.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <mutex>
#include <atomic>
#include <random>
#include <thread>

#include "model.h"
#include "rng.h"
#include "scheduler.h"

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
void sde_set_last_error(const std::string& m) { g_last_error = m; }

// ------------------------------------------------------------------------------------------ profiler
struct Profiler {
    bool enabled = false;
    struct Class { std::string name; uint64_t launches = 0; double ms = 0, flops = 0, bytes = 0; };
    std::vector<Class> classes;
    struct Pending { int cls; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    hipStream_t stream = nullptr;

    hipEvent_t get_event()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    int class_of(const char* name)
    {
        for (size_t i = 0; i < classes.size(); ++i) if (classes[i].name == name) return (int)i;
        classes.push_back(Class{name});
        return (int)classes.size() - 1;
    }
    void resolve()
    {
        if (pending.empty()) return;
        HIP_CHECK(hipStreamSynchronize(stream));
        for (auto& p : pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) classes[(size_t)p.cls].ms += ms;
            pool.push_back(p.a); pool.push_back(p.b);
        }
        pending.clear();
    }
    ~Profiler() { for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); } for (auto e : pool) (void)hipEventDestroy(e); }
};

ProfScope::ProfScope(const LaunchCtx& c, const char* name, double flops, double bytes) : ctx(c), slot(-1)
{
    Profiler* p = c.prof;
    if (!p || !p->enabled) return;
    int cls = p->class_of(name);
    p->classes[(size_t)cls].launches += 1;
    p->classes[(size_t)cls].flops += flops;
    p->classes[(size_t)cls].bytes += bytes;
    Profiler::Pending pe{cls, p->get_event(), p->get_event()};
    (void)hipEventRecord(pe.a, c.stream);
    p->pending.push_back(pe);
    slot = (int)p->pending.size() - 1;
}
ProfScope::~ProfScope()
{
    if (slot < 0) return;
    Profiler* p = ctx.prof;
    (void)hipEventRecord(p->pending[(size_t)slot].b, ctx.stream);
}

// ------------------------------------------------------------------------------------------ engine
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    void ensure(size_t bytes, bool zero = false, hipStream_t s = nullptr)
    {
        if (bytes <= cap) return;
        if (p) { HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipFree(p)); p = nullptr; cap = 0; }
        HIP_CHECK(hipMalloc(&p, bytes));
        cap = bytes;
        if (zero) HIP_CHECK(hipMemsetAsync(p, 0, bytes, s));
    }
    template <typename T> T* as() { return (T*)p; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// page-locked host staging (async copies that really are asynchronous)
struct PinnedBuf {
    void* p = nullptr; size_t cap = 0;
    void ensure(size_t bytes)
    {
        if (bytes <= cap) return;
        if (p) { HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipHostFree(p)); p = nullptr; cap = 0; }
        HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        cap = bytes;
    }
    template <typename T> T* as() { return (T*)p; }
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};

// A UNet call (fill_kv = false) captured into a hipGraph: ~440 launches become one hipGraphLaunch.  Valid for one shape and one set
// of buffer addresses; the timestep is not baked in (it is read from e->t_dev).
struct UnetGraph {
    long long key = -1;
    const void* ptrs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // x_packed, ehs, out, arena base, kv base, t_dev
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void reset()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr; graph = nullptr; key = -1;
    }
};

// A second stream with everything a VAE decode needs for itself (arena, split-K workspace, staging), so that preview
// callbacks (StableDiffusionCallback::{Latents, Decoded, ApproximateDecoded}, mod.rs:41-86) run beside the denoise stream
// instead of stalling it.  Created on first use.
struct PreviewLane {
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;   // ready: latents of the step are final (main stream) ; done: payload is in `host`
    LaunchCtx ctx;
    Arena arena;
    DevBuf arena_buf, vae_in, splitk_ws, images;
    PinnedBuf host;
    long long key = -1;
    size_t peak = 0;
    ~PreviewLane()
    {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (ready) (void)hipEventDestroy(ready);
        if (done) (void)hipEventDestroy(done);
    }
};

struct sde_engine {
    int device = 0;
    sde_engine_opts opts{};
    hipStream_t stream = nullptr;
    Profiler prof;
    LaunchCtx ctx;
    std::unique_ptr<OnnxModel> unet_onnx, vae_onnx;  // kept only during construction
    WeightBuilder wb;
    UNetModel unet;
    VaeDecoderModel vae;
    VaeEncoderModel vae_enc;
    bool has_vae_enc = false;
    std::string model_dir;          // set by sde_engine_create (sde_engine_load_vae_encoder(NULL) resolves the TOML entry against it)
    long long enc_key = -1; size_t enc_peak = 0;
    bool has_unet = false, has_vae = false;
    Arena arena;
    DevBuf arena_buf, zeros, splitk_ws, kvbuf, ehs_h, ehs_f, x_packed, model_out, latents, noise, images, vae_in, hist[4], scratch_f, taps, rescale_ws;
    PinnedBuf noise_host;
    std::unique_ptr<PreviewLane> preview;
    DevBuf t_dev;                      // the UNet call's (truncated) timestep, read by the time-embedding kernel
    std::vector<UnetGraph> graphs;     // captured UNet calls (unet_run)
    bool use_graphs = true;            // SDE_GRAPH=0: launch by launch
    int capture_fault_inject = 0;      // SDE_TEST_CAPTURE_FAULT=n: the next n graph captures fail inside (fault injection for the fallback test)
    std::string graph_fallback_reason; // why use_graphs was turned off at run time (empty: it was not)
    long long split_max_hw = 1LL << 40;   // UNet trunk as split fp16 at every level (Exec::split); SDE_SPLIT_MAX_HW: tuning hook (0 = off)
    std::vector<half_t*> kv_ptrs;
    std::mutex mu;  // one forward at a time per engine (reference: Session is Send+Sync; here calls serialize)
    std::atomic<std::thread::id> owner{std::thread::id()};  // thread inside a locked call (re-entry from a callback is refused)
    // cached arena sizing
    long long unet_key = -1, vae_key = -1;
    size_t unet_peak = 0, vae_peak = 0;

    ~sde_engine()
    {
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto& g : graphs) g.reset();
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static void engine_activate(sde_engine* e) { HIP_CHECK(hipSetDevice(e->device)); }

// Serialises calls on one engine.  Callbacks and the exchange hook run with the lock held (sdengine.h): a call back into the
// same engine from inside one is an error status, not a deadlock on the non-recursive mutex.
struct EngineLock {
    sde_engine* e;
    explicit EngineLock(sde_engine* eng) : e(eng)
    {
        SDE_REQUIRE(e->owner.load() != std::this_thread::get_id(), SDE_ERR_INVALID_ARGUMENT,
                    "engine re-entered from one of its own callbacks (callbacks must not call back into the engine that runs them)");
        e->mu.lock();
        e->owner.store(std::this_thread::get_id());
    }
    ~EngineLock() { e->owner.store(std::thread::id()); e->mu.unlock(); }
    EngineLock(const EngineLock&) = delete;
    EngineLock& operator=(const EngineLock&) = delete;
};

static void ensure_arena(sde_engine* e, size_t bytes)
{
    if (bytes > e->arena_buf.cap) e->arena_buf.ensure(bytes + (bytes >> 3));
    e->arena.base = e->arena_buf.as<char>();
    e->arena.cap = e->arena_buf.cap;
}

static void setup_kv(sde_engine* e, int N, int S)
{
    size_t total = 0;
    for (int i = 0; i < e->unet.n_transformers; ++i) total += (e->unet.kv_elems(i, N, S) + 127) & ~(size_t)127;
    e->kvbuf.ensure(total * e->ctx.esize());
    e->kv_ptrs.resize((size_t)e->unet.n_transformers);
    size_t off = 0;
    for (int i = 0; i < e->unet.n_transformers; ++i) {
        e->kv_ptrs[(size_t)i] = (half_t*)(e->kvbuf.as<char>() + off * e->ctx.esize());
        off += (e->unet.kv_elems(i, N, S) + 127) & ~(size_t)127;
    }
}

// runs the UNet on device buffers; sizes the arena with a dry pass when the shape changes
// returns true when the call went out as one hipGraph launch
static bool unet_run(sde_engine* e, const half_t* x_packed, int N, int h, int w, float t, const half_t* ehs_h, int S, bool fill_kv,
                     float* out)
{
    SDE_REQUIRE(e->has_unet, SDE_ERR_MODEL, "engine was created without a UNet");
    const long long key = ((long long)N << 40) ^ ((long long)h << 28) ^ ((long long)w << 16) ^ S;
    if (key != e->unet_key) {
        Arena dry; dry.dry = true; dry.act_bytes = e->ctx.esize();
        Exec dx{e->ctx, &dry, e->split_max_hw};
        std::vector<half_t*> fake((size_t)e->unet.n_transformers, (half_t*)nullptr);
        e->unet.forward(dx, nullptr, N, h, w, nullptr, nullptr, S, fake.data(), true, nullptr);
        e->unet_peak = dry.peak; e->unet_key = key;
    }
    ensure_arena(e, std::max(e->unet_peak, e->vae_peak));
    e->t_dev.ensure(256);
    // the exported graph casts the timestep to int64 (sd2pyke.py:121): truncation
    launch_set_scalar(e->ctx, e->t_dev.as<float>(), truncf(t));
    auto run_plain = [&] {
        e->arena.off = 0; e->arena.peak = 0; e->arena.dry = false; e->arena.act_bytes = e->ctx.esize();
        Exec ex{e->ctx, &e->arena, e->split_max_hw};
        e->unet.forward(ex, x_packed, N, h, w, e->t_dev.as<float>(), ehs_h, S, e->kv_ptrs.data(), fill_kv, out);
    };
    // the first call of a run (fill_kv) and profiled passes (events between the launches) go out launch by launch
    if (!e->use_graphs || fill_kv || e->prof.enabled) { run_plain(); return false; }
    const void* ptrs[6] = {x_packed, ehs_h, out, e->arena_buf.p, e->kvbuf.p, e->t_dev.p};
    UnetGraph* g = nullptr;
    for (auto& c : e->graphs)
        if (c.key == key && std::equal(ptrs, ptrs + 6, c.ptrs)) g = &c;
    if (!g) {
        // capture (nothing executes here).  Every kernel's one-time set-up (dynamic-LDS attributes) was done by the uncaptured call
        // that filled the K/V cache for this shape.
        UnetGraph ng;
        ng.key = key; std::copy(ptrs, ptrs + 6, ng.ptrs);
        hipError_t err = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
        bool threw = false;
        std::string why;
        if (err == hipSuccess) {
            // Anything run_plain() throws while the stream is capturing (a HIP_CHECK on a capture-illegal call: a kernel instance's
            // first-time hipFuncSetAttribute, a DevBuf::ensure that has to grow ...) must not reach the caller with the stream still
            // capturing and use_graphs still set: every later call of this shape would fail the same way.
            try {
                if (e->capture_fault_inject > 0) { --e->capture_fault_inject; throw SdeError(SDE_ERR_DEVICE, "injected capture fault (SDE_TEST_CAPTURE_FAULT)"); }
                run_plain();
            } catch (const std::exception& ex) { threw = true; why = ex.what(); } catch (...) { threw = true; why = "unknown exception"; }
            hipError_t end = hipStreamEndCapture(e->stream, &ng.graph);
            if (threw) err = hipErrorStreamCaptureInvalidated;
            else err = end;
            if (err == hipSuccess) err = hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0);
        }
        if (err != hipSuccess) {   // no graph for this call: stay on plain launches from here on
            (void)hipGetLastError();
            ng.reset();
            e->use_graphs = false;
            e->graph_fallback_reason = threw ? why : std::string(hipGetErrorString(err));
            run_plain();           // uncaptured replay; if THIS throws the error is real and goes to the caller
            return false;
        }
        if (e->graphs.size() >= 4) {
            // the oldest graph may still be executing (launched by an earlier step of this stream)
            HIP_CHECK(hipStreamSynchronize(e->stream));
            e->graphs.front().reset(); e->graphs.erase(e->graphs.begin());
        }
        e->graphs.push_back(ng);
        g = &e->graphs.back();
    }
    HIP_CHECK(hipGraphLaunch(g->exec, e->stream));
    return true;
}

static void vae_run(sde_engine* e, const half_t* x_packed, int B, int h, int w, int epi, float* out)
{
    SDE_REQUIRE(e->has_vae, SDE_ERR_MODEL, "engine was created without a VAE decoder");
    const long long key = ((long long)B << 40) ^ ((long long)h << 20) ^ w;
    if (key != e->vae_key) {
        Arena dry; dry.dry = true; dry.act_bytes = e->ctx.esize();
        Exec dx{e->ctx, &dry};
        e->vae.forward(dx, nullptr, B, h, w, epi, nullptr);
        e->vae_peak = dry.peak; e->vae_key = key;
    }
    ensure_arena(e, std::max(e->unet_peak, e->vae_peak));
    e->arena.off = 0; e->arena.peak = 0; e->arena.dry = false; e->arena.act_bytes = e->ctx.esize();
    Exec ex{e->ctx, &e->arena};
    e->vae.forward(ex, x_packed, B, h, w, epi, out);
}

// decode_latents on device: latents f32 NCHW (device) -> images f32 NHWC (device)
static void decode_dev(sde_engine* e, const float* d_latents, int B, int h, int w, int epi, float* d_out)
{
    // impl_main.rs:399 — `1.0 / 0.18215 * latents`; the TOML scale-factor is ignored there too
    const float scale = epi == EPI_IMAGE ? 1.0f / 0.18215f : 1.0f;
    // the reference decodes one image per ORT call (impl_main.rs:402-405); batching is numerically identical per image.
    // Chunk so the 512^2 activations stay within a few GB.
    const int chunk = std::max(1, std::min(B, 8));
    const int up = e->vae.upscale();
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        e->vae_in.ensure((size_t)nb * h * w * VaeDecoderModel::kInPad * e->ctx.esize());
        launch_pack_nchw(e->ctx, d_latents + (size_t)b0 * e->vae.latent_channels * h * w, e->vae_in.as<half_t>(), nb, e->vae.latent_channels, h, w,
                         VaeDecoderModel::kInPad, scale);
        vae_run(e, e->vae_in.as<half_t>(), nb, h, w, epi, d_out + (size_t)b0 * e->vae.out_channels * h * up * w * up);
    }
}

static PreviewLane& preview_lane(sde_engine* e)
{
    if (!e->preview) {
        std::unique_ptr<PreviewLane> L(new PreviewLane());
        HIP_CHECK(hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&L->ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&L->done, hipEventDisableTiming));
        L->ctx = e->ctx; L->ctx.stream = L->stream; L->ctx.prof = nullptr;
        L->splitk_ws.ensure((size_t)64 << 20);
        L->ctx.splitk_ws = L->splitk_ws.as<float>(); L->ctx.splitk_ws_bytes = L->splitk_ws.cap;
        e->preview = std::move(L);
    }
    return *e->preview;
}

// decode_latents on the preview lane (its own stream, arena and workspaces): same kernels, same arithmetic as decode_dev
static void decode_on_lane(sde_engine* e, PreviewLane& L, const float* d_latents, int B, int h, int w, float* d_out)
{
    SDE_REQUIRE(e->has_vae, SDE_ERR_MODEL, "Decoded callback needs the VAE decoder");
    const int chunk = std::max(1, std::min(B, 8)), up = e->vae.upscale();
    const long long key = ((long long)chunk << 40) ^ ((long long)h << 20) ^ w;
    if (key != L.key) {
        Arena dry; dry.dry = true; dry.act_bytes = e->ctx.esize();
        Exec dx{L.ctx, &dry};
        e->vae.forward(dx, nullptr, chunk, h, w, EPI_IMAGE, nullptr);
        L.peak = dry.peak; L.key = key;
    }
    if (L.peak > L.arena_buf.cap) { HIP_CHECK(hipStreamSynchronize(L.stream)); L.arena_buf.ensure(L.peak + (L.peak >> 3)); }
    L.arena.base = L.arena_buf.as<char>(); L.arena.cap = L.arena_buf.cap; L.arena.dry = false; L.arena.act_bytes = e->ctx.esize();
    L.vae_in.ensure((size_t)chunk * h * w * VaeDecoderModel::kInPad * e->ctx.esize());
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        launch_pack_nchw(L.ctx, d_latents + (size_t)b0 * e->vae.latent_channels * h * w, L.vae_in.as<half_t>(), nb, e->vae.latent_channels, h, w,
                         VaeDecoderModel::kInPad, 1.0f / 0.18215f);
        L.arena.off = 0; L.arena.peak = 0;
        Exec ex{L.ctx, &L.arena};
        e->vae.forward(ex, L.vae_in.as<half_t>(), nb, h, w, EPI_IMAGE, d_out + (size_t)b0 * e->vae.out_channels * h * up * w * up);
    }
}

// ------------------------------------------------------------------------------------------ static scratch (no-allocation entry point)
// Staging memory of sde_approximate_decode_noalloc: a file-scope device array, so the call needs no hipMalloc.  Every GPU this
// library runs on holds its own copy (SDE_NOALLOC_SCRATCH_BYTES, until the process ends) at its own address; in host code the
// array's NAME is only a handle for hipGetSymbolAddress -- it is never passed to a kernel or a copy.  The per-device address is
// looked up once, with that device current, and kept in g_scratch_ptr.
static __device__ __align__(256) unsigned char g_noalloc_scratch[SDE_NOALLOC_SCRATCH_BYTES];
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;                       // guards the table below
void* g_scratch_ptr[kMaxDevices] = {nullptr};
std::mutex g_scratch_use_mu[kMaxDevices];      // one user of a device's array at a time inside this process

// `device` must be the current device
void* noalloc_scratch_for(int device)
{
    SDE_REQUIRE(device >= 0 && device < kMaxDevices, SDE_ERR_DEVICE, "device ordinal beyond the static scratch table");
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch)));
        SDE_REQUIRE(p != nullptr, SDE_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the static scratch array");
        g_scratch_ptr[device] = p;
    }
    return g_scratch_ptr[device];
}
}  // namespace

// ------------------------------------------------------------------------------------------ C ABI: library / engine
extern "C" {

const char* sde_last_error(void) { return g_last_error.c_str(); }
const char* sde_version(void) { return "sdengine 0.1 (gfx950)"; }
int sde_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void sde_engine_opts_default(sde_engine_opts* o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->device_id = 0; o->dtype = SDE_F16; o->force_naive = 0; o->load_unet = 1; o->load_vae = 1; o->num_heads_override = 0;
}

static sde_status engine_create_impl(const std::string& unet_path, const std::string& vae_path, const sde_engine_opts* opts_in, sde_engine** out)
{
    return guarded([&] {
        SDE_REQUIRE(out, SDE_ERR_INVALID_ARGUMENT, "out pointer is null");
        *out = nullptr;
        sde_engine_opts o;
        if (opts_in) o = *opts_in; else sde_engine_opts_default(&o);
        SDE_REQUIRE(o.dtype == SDE_F16 || o.dtype == SDE_F32, SDE_ERR_INVALID_ARGUMENT,
                    "engine dtype must be f16 (MFMA kernels, the benchmarked path) or f32 (plain fp32 kernels, the reference's own precision); bf16 is not implemented");
        int ndev = sde_device_count();
        SDE_REQUIRE(ndev > 0, SDE_ERR_DEVICE, "no HIP device visible: the engine has no CPU fallback");
        SDE_REQUIRE(o.device_id >= 0 && o.device_id < ndev, SDE_ERR_DEVICE, "device_id out of range");
        std::unique_ptr<sde_engine> e(new sde_engine());
        e->device = o.device_id; e->opts = o;
        HIP_CHECK(hipSetDevice(e->device));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, e->device));
        SDE_REQUIRE(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0, SDE_ERR_DEVICE,
                    std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName);
        HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        (void)noalloc_scratch_for(e->device);   // resolve this device's static scratch address at set-up (sde_approximate_decode_noalloc)
        e->prof.stream = e->stream;
        e->zeros.ensure(4096, true, e->stream);
        e->ctx.stream = e->stream; e->ctx.prof = &e->prof; e->ctx.zeros = e->zeros.as<half_t>(); e->ctx.force_naive = o.force_naive != 0;
        e->ctx.f32 = o.dtype == SDE_F32;
        if (e->ctx.f32) e->split_max_hw = 0;   // floats need no low parts
        e->splitk_ws.ensure((size_t)64 << 20);
        e->ctx.splitk_ws = e->splitk_ws.as<float>(); e->ctx.splitk_ws_bytes = e->splitk_ws.cap;
        if (const char* sk = getenv("SDE_SPLITK")) e->ctx.splitk_force = atoi(sk);   // tuning hook (0 = heuristic, 1 = off)
        if (const char* sp = getenv("SDE_SPLIT_MAX_HW")) { if (!e->ctx.f32) e->split_max_hw = atoll(sp); }
        if (const char* gr = getenv("SDE_GRAPH")) e->use_graphs = atoi(gr) != 0;
        if (const char* cf = getenv("SDE_TEST_CAPTURE_FAULT")) e->capture_fault_inject = atoi(cf);
        e->wb.ctx = e->ctx; e->wb.ctx.prof = nullptr;
        if (o.load_unet && !unet_path.empty()) {
            OnnxModel m; WeightMap w;
            onnx_load(unet_path, m);
            bind_weights(m, w);
            e->wb.w = &w; e->wb.params = 0; e->wb.gelu_tanh = m.fast_gelu;
            e->unet.build(m, w, e->wb, o.num_heads_override);
            HIP_CHECK(hipStreamSynchronize(e->stream));  // repack kernels read mapped files' staging copies
            e->has_unet = true;
        }
        if (o.load_vae && !vae_path.empty()) {
            OnnxModel m; WeightMap w;
            onnx_load(vae_path, m);
            bind_weights(m, w);
            e->wb.w = &w; e->wb.params = 0;
            e->vae.build(m, w, e->wb);
            HIP_CHECK(hipStreamSynchronize(e->stream));
            e->has_vae = true;
        }
        e->wb.w = nullptr;
        e->wb.release_staging();
        SDE_REQUIRE(e->has_unet || e->has_vae, SDE_ERR_INVALID_ARGUMENT, "nothing to load");
        *out = e.release();
    });
}

sde_status sde_engine_create(const char* model_dir, const sde_engine_opts* opts, sde_engine** out)
{
    std::string up, vp;
    sde_status st = guarded([&] {
        SDE_REQUIRE(model_dir, SDE_ERR_INVALID_ARGUMENT, "model_dir is null");
        PykeConfig cfg;
        pyke_config_load(model_dir, cfg);
        up = std::string(model_dir) + "/" + cfg.unet_path;
        vp = std::string(model_dir) + "/" + cfg.vae_decoder_path;
    });
    if (st != SDE_OK) return st;
    st = engine_create_impl(up, vp, opts, out);
    if (st == SDE_OK && out && *out) (*out)->model_dir = model_dir;
    return st;
}

sde_status sde_engine_create_from_files(const char* unet_onnx, const char* vae_decoder_onnx, const sde_engine_opts* opts, sde_engine** out)
{
    return engine_create_impl(unet_onnx ? unet_onnx : "", vae_decoder_onnx ? vae_decoder_onnx : "", opts, out);
}

void sde_engine_destroy(sde_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    delete e;
}

sde_status sde_engine_info(const sde_engine* e, sde_model_info* info)
{
    return guarded([&] {
        SDE_REQUIRE(e && info, SDE_ERR_INVALID_ARGUMENT, "null argument");
        memset(info, 0, sizeof(*info));
        info->unet_loaded = e->has_unet; info->vae_loaded = e->has_vae;
        if (e->has_unet) {
            const UNetModel& u = e->unet;
            info->in_channels = u.in_channels; info->out_channels = u.out_channels; info->n_levels = u.n_levels;
            info->layers_per_block = u.layers_per_block; info->cross_attention_dim = u.ctx_dim; info->use_linear_projection = u.linear_proj;
            for (int i = 0; i < u.n_levels && i < 8; ++i) { info->block_out_channels[i] = u.boc[(size_t)i]; info->num_heads[i] = u.heads[(size_t)i]; info->down_attn[i] = u.down_attn[(size_t)i]; }
            info->unet_params = u.n_params;
        }
        if (e->has_vae) {
            const VaeDecoderModel& v = e->vae;
            info->vae_n_levels = v.n_levels; info->vae_layers_per_block = v.layers_per_block; info->vae_latent_channels = v.latent_channels;
            info->vae_out_channels = v.out_channels;
            for (int i = 0; i < v.n_levels && i < 8; ++i) info->vae_block_out_channels[i] = v.boc[(size_t)i];
            info->vae_params = v.n_params;
        }
    });
}

// ------------------------------------------------------------------------------------------ C ABI: UNet / VAE calls
static void check_unet_args(const sde_engine* e, const void* sample, int N, int h, int w, const void* ehs, int S, int D, const void* out)
{
    SDE_REQUIRE(e && sample && ehs && out, SDE_ERR_INVALID_ARGUMENT, "null argument");
    SDE_REQUIRE(e->has_unet, SDE_ERR_MODEL, "engine was created without a UNet");
    SDE_REQUIRE(N > 0 && h > 0 && w > 0 && S > 0, SDE_ERR_INVALID_ARGUMENT, "empty tensor");
    SDE_REQUIRE(D == e->unet.ctx_dim, SDE_ERR_INVALID_ARGUMENT,
                "encoder_hidden_states width " + std::to_string(D) + " != cross_attention_dim " + std::to_string(e->unet.ctx_dim));
}

// caller holds the engine lock
static void unet_forward_dev_locked(sde_engine* e, const float* d_sample, int N, int h, int w, float timestep, const float* d_ehs, int S, int D,
                                    float* d_out)
{
    const int Cp = UNetModel::kInPad;
    e->x_packed.ensure((size_t)N * h * w * Cp * e->ctx.esize());
    e->ehs_h.ensure((size_t)N * S * D * e->ctx.esize());
    setup_kv(e, N, S);
    launch_pack_nchw(e->ctx, d_sample, e->x_packed.as<half_t>(), N, e->unet.in_channels, h, w, Cp, 1.0f);
    launch_f32_to_f16(e->ctx, d_ehs, e->ehs_h.as<half_t>(), (int64_t)N * S * D);
    unet_run(e, e->x_packed.as<half_t>(), N, h, w, timestep, e->ehs_h.as<half_t>(), S, true, d_out);
    HIP_CHECK(hipStreamSynchronize(e->stream));
}

sde_status sde_unet_forward_dev(sde_engine* e, const float* d_sample, int N, int h, int w, float timestep, const float* d_ehs, int S, int D,
                                float* d_out)
{
    return guarded([&] {
        check_unet_args(e, d_sample, N, h, w, d_ehs, S, D, d_out);
        EngineLock lk(e);
        engine_activate(e);
        unet_forward_dev_locked(e, d_sample, N, h, w, timestep, d_ehs, S, D, d_out);
    });
}

sde_status sde_unet_forward(sde_engine* e, const float* sample, int N, int h, int w, float timestep, const float* ehs, int S, int D, float* out)
{
    return guarded([&] {
        check_unet_args(e, sample, N, h, w, ehs, S, D, out);
        EngineLock lk(e);   // held across upload, forward and download: the staging buffer is this call's alone
        engine_activate(e);
        const size_t ns = (size_t)N * e->unet.in_channels * h * w, ne = (size_t)N * S * D, no = (size_t)N * e->unet.out_channels * h * w;
        e->scratch_f.ensure((ns + ne + no) * sizeof(float));
        float* d = e->scratch_f.as<float>();
        HIP_CHECK(hipMemcpy(d, sample, ns * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d + ns, ehs, ne * 4, hipMemcpyHostToDevice));
        unet_forward_dev_locked(e, d, N, h, w, timestep, d + ns, S, D, d + ns + ne);
        HIP_CHECK(hipMemcpy(out, d + ns + ne, no * 4, hipMemcpyDeviceToHost));
    });
}

static sde_status vae_common(sde_engine* e, const float* latents, int B, int h, int w, float* out, int epi, bool dev)
{
    return guarded([&] {
        SDE_REQUIRE(e && latents && out, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(e->has_vae, SDE_ERR_MODEL, "engine was created without a VAE decoder");
        SDE_REQUIRE(B > 0 && h > 0 && w > 0, SDE_ERR_INVALID_ARGUMENT, "empty tensor");
        EngineLock lk(e);
        engine_activate(e);
        const int up = e->vae.upscale();
        const size_t nl = (size_t)B * e->vae.latent_channels * h * w, no = (size_t)B * e->vae.out_channels * h * up * w * up;
        if (dev) {
            decode_dev(e, latents, B, h, w, epi, out);
            HIP_CHECK(hipStreamSynchronize(e->stream));
        } else {
            e->latents.ensure(nl * 4);
            e->images.ensure(no * 4);
            HIP_CHECK(hipMemcpy(e->latents.p, latents, nl * 4, hipMemcpyHostToDevice));
            decode_dev(e, e->latents.as<float>(), B, h, w, epi, e->images.as<float>());
            HIP_CHECK(hipStreamSynchronize(e->stream));
            HIP_CHECK(hipMemcpy(out, e->images.p, no * 4, hipMemcpyDeviceToHost));
        }
    });
}
sde_status sde_vae_decode(sde_engine* e, const float* latents, int B, int h, int w, float* images) { return vae_common(e, latents, B, h, w, images, EPI_IMAGE, false); }
sde_status sde_vae_decode_dev(sde_engine* e, const float* latents, int B, int h, int w, float* images) { return vae_common(e, latents, B, h, w, images, EPI_IMAGE, true); }
sde_status sde_vae_decoder_run(sde_engine* e, const float* latents, int B, int h, int w, float* sample) { return vae_common(e, latents, B, h, w, sample, EPI_NCHW_F32, false); }

sde_status sde_engine_load_vae_encoder(sde_engine* e, const char* onnx_path)
{
    return guarded([&] {
        SDE_REQUIRE(e, SDE_ERR_INVALID_ARGUMENT, "null engine");
        std::string path = onnx_path ? onnx_path : "";
        if (path.empty()) {
            SDE_REQUIRE(!e->model_dir.empty(), SDE_ERR_INVALID_ARGUMENT, "no path given and the engine was not created from a model directory");
            PykeConfig cfg;
            pyke_config_load(e->model_dir, cfg);
            SDE_REQUIRE(!cfg.vae_encoder_path.empty(), SDE_ERR_MODEL, "pyke-diffusers.toml names no [vae] encoder");
            path = e->model_dir + "/" + cfg.vae_encoder_path;
        }
        EngineLock lk(e);
        engine_activate(e);
        OnnxModel m; WeightMap w;
        onnx_load(path, m);
        bind_weights(m, w);
        e->wb.w = &w; e->wb.params = 0; e->wb.gelu_tanh = false;
        e->vae_enc = VaeEncoderModel();
        e->has_vae_enc = false;
        e->vae_enc.build(m, w, e->wb);
        HIP_CHECK(hipStreamSynchronize(e->stream));
        e->wb.w = nullptr;
        e->wb.release_staging();
        e->has_vae_enc = true; e->enc_key = -1;
    });
}

sde_status sde_vae_encoder_out_shape(const sde_engine* e, int H, int W, int* latent_channels, int* h, int* w)
{
    return guarded([&] {
        SDE_REQUIRE(e && latent_channels && h && w, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(e->has_vae_enc, SDE_ERR_MODEL, "no VAE encoder loaded (sde_engine_load_vae_encoder)");
        SDE_REQUIRE(H >= e->vae_enc.downscale() && W >= e->vae_enc.downscale(), SDE_ERR_INVALID_ARGUMENT, "image smaller than the encoder's downscale factor");
        *latent_channels = e->vae_enc.latent_channels;
        *h = VaeEncoderModel::out_size(H, e->vae_enc.n_levels); *w = VaeEncoderModel::out_size(W, e->vae_enc.n_levels);
    });
}

sde_status sde_vae_encoder_run(sde_engine* e, const float* sample, int B, int H, int W, const float* noise, float* latent_sample)
{
    return guarded([&] {
        SDE_REQUIRE(e && sample && latent_sample, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(e->has_vae_enc, SDE_ERR_MODEL, "no VAE encoder loaded (sde_engine_load_vae_encoder)");
        SDE_REQUIRE(B > 0 && H > 0 && W > 0, SDE_ERR_INVALID_ARGUMENT, "empty tensor");
        const VaeEncoderModel& v = e->vae_enc;
        SDE_REQUIRE(H >= v.downscale() && W >= v.downscale(), SDE_ERR_INVALID_ARGUMENT, "image smaller than the encoder's downscale factor");
        EngineLock lk(e);
        engine_activate(e);
        const int h = VaeEncoderModel::out_size(H, v.n_levels), w = VaeEncoderModel::out_size(W, v.n_levels), LC = v.latent_channels;
        const size_t ni = (size_t)B * v.in_channels * H * W, nl = (size_t)B * LC * h * w;
        // staging: image f32 | noise f32 | moments f32 | sample f32, then the packed fp16 image
        e->scratch_f.ensure((ni + nl + 2 * nl + nl) * sizeof(float));
        float* d_img = e->scratch_f.as<float>();
        float* d_noise = d_img + ni; float* d_mom = d_noise + nl; float* d_out = d_mom + 2 * nl;
        HIP_CHECK(hipMemcpyAsync(d_img, sample, ni * 4, hipMemcpyHostToDevice, e->stream));
        if (noise) HIP_CHECK(hipMemcpyAsync(d_noise, noise, nl * 4, hipMemcpyHostToDevice, e->stream));
        e->vae_in.ensure((size_t)B * H * W * VaeEncoderModel::kInPad * e->ctx.esize());
        launch_pack_nchw(e->ctx, d_img, e->vae_in.as<half_t>(), B, v.in_channels, H, W, VaeEncoderModel::kInPad, 1.0f);
        const long long key = ((long long)B << 40) ^ ((long long)H << 20) ^ W;
        if (key != e->enc_key) {
            Arena dry; dry.dry = true; dry.act_bytes = e->ctx.esize();
            Exec dx{e->ctx, &dry};
            v.forward(dx, nullptr, B, H, W, nullptr);
            e->enc_peak = dry.peak; e->enc_key = key;
        }
        ensure_arena(e, std::max(std::max(e->unet_peak, e->vae_peak), e->enc_peak));
        e->arena.off = 0; e->arena.peak = 0; e->arena.dry = false; e->arena.act_bytes = e->ctx.esize();
        Exec ex{e->ctx, &e->arena};
        v.forward(ex, e->vae_in.as<half_t>(), B, H, W, d_mom);
        launch_vae_sample(e->ctx, d_mom, noise ? d_noise : nullptr, d_out, B, LC, h, w);
        HIP_CHECK(hipMemcpyAsync(latent_sample, d_out, nl * 4, hipMemcpyDeviceToHost, e->stream));
        HIP_CHECK(hipStreamSynchronize(e->stream));
    });
}

sde_status sde_approximate_decode(sde_engine* e, const float* latents, int B, int h, int w, float* images)
{
    return guarded([&] {
        SDE_REQUIRE(e && latents && images && B > 0 && h > 0 && w > 0, SDE_ERR_INVALID_ARGUMENT, "bad argument");
        EngineLock lk(e);
        engine_activate(e);
        const size_t nl = (size_t)B * 4 * h * w, no = (size_t)B * h * w * 3;
        e->latents.ensure(nl * 4); e->images.ensure(no * 4);
        HIP_CHECK(hipMemcpy(e->latents.p, latents, nl * 4, hipMemcpyHostToDevice));
        launch_approx_decode(e->ctx, e->latents.as<float>(), e->images.as<float>(), B, h, w);
        HIP_CHECK(hipStreamSynchronize(e->stream));
        HIP_CHECK(hipMemcpy(images, e->images.p, no * 4, hipMemcpyDeviceToHost));
    });
}

sde_status sde_approximate_decode_noalloc(sde_engine* e, const float* latents, int B, int h, int w, float* images)
{
    return guarded([&] {
        SDE_REQUIRE(e && latents && images && B > 0 && h > 0 && w > 0, SDE_ERR_INVALID_ARGUMENT, "bad argument");
        const size_t nl = (size_t)B * 4 * h * w, no = (size_t)B * h * w * 3;
        const size_t off_out = (nl * 4 + 255) & ~(size_t)255;
        SDE_REQUIRE(off_out + no * 4 <= (size_t)SDE_NOALLOC_SCRATCH_BYTES, SDE_ERR_INVALID_ARGUMENT,
                    "sde_approximate_decode_noalloc: B*h*w = " + std::to_string((size_t)B * h * w) + " pixels need " +
                        std::to_string(off_out + no * 4) + " bytes of staging, the static scratch holds " +
                        std::to_string((size_t)SDE_NOALLOC_SCRATCH_BYTES) + " (use sde_approximate_decode)");
        EngineLock lk(e);
        engine_activate(e);
        std::lock_guard<std::mutex> use(g_scratch_use_mu[e->device]);   // other engines on this GPU share the array
        unsigned char* base = (unsigned char*)noalloc_scratch_for(e->device);
        float* d_lat = (float*)base;
        float* d_img = (float*)(base + off_out);
        // nothing has to start at zero: the kernel reads [0, nl) of what this call uploads and writes all of [0, no)
        HIP_CHECK(hipMemcpyAsync(d_lat, latents, nl * 4, hipMemcpyHostToDevice, e->stream));
        launch_approx_decode(e->ctx, d_lat, d_img, B, h, w);
        HIP_CHECK(hipMemcpyAsync(images, d_img, no * 4, hipMemcpyDeviceToHost, e->stream));
        HIP_CHECK(hipStreamSynchronize(e->stream));
    });
}

// ------------------------------------------------------------------------------------------ C ABI: schedulers
void sde_sched_cfg_default(int kind, sde_sched_cfg* c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->num_train_timesteps = 1000; c->beta_start = 0.0001f; c->beta_end = 0.02f; c->beta_schedule = SDE_BETA_LINEAR;
    c->prediction_type = SDE_PRED_EPSILON; c->reference_quirks = 1;
    c->solver_order = 2; c->lower_order_final = 1;                 // dpm_solver_multistep.rs:80-92
    c->clip_sample = kind == SDE_SCHED_DDIM ? 1 : 0;               // ddim.rs:59-67 ; ddpm.rs:40 (derive Default => false)
    c->set_alpha_to_one = 1; c->steps_offset = 0; c->variance_type = 0;
    c->algorithm_type = SDE_DPM_SOLVER_PP; c->solver_type = SDE_DPM_MIDPOINT;   // dpm_solver_multistep.rs:87-88
}
void sde_sched_cfg_sd_v1(int kind, sde_sched_cfg* c)
{
    if (!c) return;
    sde_sched_cfg_default(kind, c);
    c->beta_start = 0.00085f; c->beta_end = 0.012f; c->beta_schedule = SDE_BETA_SCALED_LINEAR;
    if (kind == SDE_SCHED_DDIM) { c->clip_sample = 0; c->set_alpha_to_one = 0; c->steps_offset = 1; }  // ddim.rs:275-286
}

sde_status sde_scheduler_create(int kind, const sde_sched_cfg* cfg, sde_scheduler** out)
{
    return guarded([&] {
        SDE_REQUIRE(out, SDE_ERR_INVALID_ARGUMENT, "out pointer is null");
        *out = nullptr;
        sde_sched_cfg c;
        if (cfg) c = *cfg; else sde_sched_cfg_sd_v1(kind, &c);
        std::unique_ptr<sde_scheduler> s(new sde_scheduler());
        s->init(kind, c);
        *out = s.release();
    });
}
void sde_scheduler_destroy(sde_scheduler* s) { delete s; }
int sde_scheduler_order(const sde_scheduler*) { return 1; }  // every scheduler in the reference returns 1
int sde_scheduler_len(const sde_scheduler* s) { return s ? s->T : 0; }
sde_status sde_scheduler_set_timesteps(sde_scheduler* s, int n)
{
    return guarded([&] { SDE_REQUIRE(s, SDE_ERR_INVALID_ARGUMENT, "null scheduler"); s->set_timesteps(n); });
}
int sde_scheduler_num_timesteps(const sde_scheduler* s) { return s ? (int)s->timesteps.size() : 0; }
sde_status sde_scheduler_timesteps(const sde_scheduler* s, float* out, int capacity)
{
    return guarded([&] {
        SDE_REQUIRE(s && out, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(capacity >= (int)s->timesteps.size(), SDE_ERR_INVALID_ARGUMENT, "capacity too small");
        std::copy(s->timesteps.begin(), s->timesteps.end(), out);
    });
}
sde_status sde_scheduler_sigmas(const sde_scheduler* s, float* out, int capacity, int* n_out)
{
    return guarded([&] {
        SDE_REQUIRE(s && n_out, SDE_ERR_INVALID_ARGUMENT, "null argument");
        *n_out = (int)s->sigmas.size();
        if (out) { SDE_REQUIRE(capacity >= *n_out, SDE_ERR_INVALID_ARGUMENT, "capacity too small"); std::copy(s->sigmas.begin(), s->sigmas.end(), out); }
    });
}
float sde_scheduler_init_noise_sigma(const sde_scheduler* s) { return s ? s->init_noise_sigma : 0.f; }

// Host-pointer mirrors of the tensor methods: they run the SAME device kernels as the fused loop (on the
// current device's null stream) — there is no CPU implementation of the step arithmetic in the product.
namespace {
struct TmpDev {
    std::vector<void*> ptrs;
    float* up(const float* h, size_t n) { float* d = nullptr; HIP_CHECK(hipMalloc((void**)&d, n * 4)); ptrs.push_back(d); if (h) HIP_CHECK(hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice)); return d; }
    ~TmpDev() { for (void* p : ptrs) (void)hipFree(p); }
};
void require_device() { SDE_REQUIRE(sde_device_count() > 0, SDE_ERR_DEVICE, "no HIP device visible: scheduler tensor methods run on the GPU"); }
}  // namespace

sde_status sde_scheduler_scale_model_input(sde_scheduler* s, const float* sample, size_t n, float timestep, float* out)
{
    return guarded([&] {
        SDE_REQUIRE(s && sample && out && n > 0, SDE_ERR_INVALID_ARGUMENT, "bad argument");
        require_device();
        float div = s->input_scale_divisor(timestep);
        TmpDev t; LaunchCtx ctx;
        float* dx = t.up(sample, n); float* dy = t.up(nullptr, n);
        launch_vec_op(ctx, dx, nullptr, dy, (int64_t)n, 0, div, 0.f);
        HIP_CHECK(hipMemcpy(out, dy, n * 4, hipMemcpyDeviceToHost));
    });
}

static void scheduler_step_impl(sde_scheduler* s, const float* model_output, float timestep, const float* sample, const float* noise, size_t n,
                                sde_step_output* o)
{
    SDE_REQUIRE(s && model_output && sample && o && o->prev_sample && n > 0, SDE_ERR_INVALID_ARGUMENT, "bad argument");
    require_device();
    StepPlan p;
    s->plan(timestep, p);
    SDE_REQUIRE(!p.needs_noise || noise, SDE_ERR_INVALID_ARGUMENT, "this scheduler step needs a noise tensor");
    // which Option<> fields of SchedulerStepOutput the step fills: pred_original_sample by Euler, Euler-a, DDIM, DDPM
    // (euler_discrete.rs:216-220, euler_ancestral_discrete.rs:206-210, ddim.rs:246-250, ddpm.rs:234-238), none by DPM-Solver
    // (dpm_solver_multistep.rs:379); `derivative` is computed by the Euler family (euler_discrete.rs:210) and on offer here
    const bool has_x0 = p.kind == STEP_EULER || p.kind == STEP_EULER_A || p.kind == STEP_DDIM || p.kind == STEP_DDPM || p.kind == STEP_LMS;
    const bool has_d = p.kind == STEP_EULER || p.kind == STEP_EULER_A || p.kind == STEP_LMS;
    TmpDev t; LaunchCtx ctx;
    StepArgs a;
    a.kind = p.kind; a.pred = p.pred; a.B = 1; a.C = 1; a.H = 1; a.W = (int)n; a.cfg = 0;
    a.model_out = t.up(model_output, n); a.latents = t.up(sample, n); a.noise = p.needs_noise ? t.up(noise, n) : nullptr;
    if (o->pred_original_sample && has_x0) a.x0_out = t.up(nullptr, n);
    if (o->derivative && has_d) a.deriv_out = t.up(nullptr, n);
    for (int i = 0; i < 12; ++i) a.c[i] = p.c[i];
    a.clip = p.clip; a.nhist = p.nhist; a.dpm_algo = p.dpm_algo;
    for (int i = 0; i < 4; ++i) a.lms_c[i] = p.lms_c[i];
    if (p.nhist > 0 || s->kind == SDE_SCHED_LMS) {
        if (s->mirror_hist_n != n) {
            for (float* q : s->mirror_hist) if (q) (void)hipFree(q);
            s->mirror_hist.assign(4, nullptr);
            for (auto& q : s->mirror_hist) { HIP_CHECK(hipMalloc((void**)&q, n * 4)); HIP_CHECK(hipMemset(q, 0, n * 4)); }
            s->mirror_hist_n = n;
        }
        for (int i = 0; i < 4; ++i) a.hist[i] = s->mirror_hist[(size_t)i];
    }
    launch_cfg_step(ctx, a);
    HIP_CHECK(hipMemcpy(o->prev_sample, a.latents, n * 4, hipMemcpyDeviceToHost));
    if (a.x0_out) HIP_CHECK(hipMemcpy(o->pred_original_sample, a.x0_out, n * 4, hipMemcpyDeviceToHost));
    if (a.deriv_out) HIP_CHECK(hipMemcpy(o->derivative, a.deriv_out, n * 4, hipMemcpyDeviceToHost));
    o->has_pred_original_sample = has_x0 ? 1 : 0;
    o->has_derivative = has_d ? 1 : 0;
    if (!s->mirror_hist.empty()) s->rotate_history(p, s->mirror_hist.data());
    s->advance(p);
}

sde_status sde_scheduler_step(sde_scheduler* s, const float* model_output, float timestep, const float* sample, const float* noise, size_t n,
                              float* prev_sample)
{
    return guarded([&] {
        sde_step_output o{};
        o.prev_sample = prev_sample;
        scheduler_step_impl(s, model_output, timestep, sample, noise, n, &o);
    });
}

sde_status sde_scheduler_step_output(sde_scheduler* s, const float* model_output, float timestep, const float* sample, const float* noise, size_t n,
                                     sde_step_output* out)
{
    return guarded([&] { scheduler_step_impl(s, model_output, timestep, sample, noise, n, out); });
}

sde_status sde_scheduler_add_noise(sde_scheduler* s, const float* original, const float* noise, size_t n, float timestep, float* out)
{
    return guarded([&] {
        SDE_REQUIRE(s && original && noise && out && n > 0, SDE_ERR_INVALID_ARGUMENT, "bad argument");
        require_device();
        float a, b;
        s->add_noise_coeffs(timestep, a, b);
        TmpDev t; LaunchCtx ctx;
        float* dx = t.up(original, n); float* dn = t.up(noise, n); float* dy = t.up(nullptr, n);
        launch_vec_op(ctx, dx, dn, dy, (int64_t)n, 1, a, b);
        HIP_CHECK(hipMemcpy(out, dy, n * 4, hipMemcpyDeviceToHost));
    });
}

// ------------------------------------------------------------------------------------------ C ABI: txt2img
void sde_txt2img_opts_default(sde_txt2img_opts* o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->height = 512; o->width = 512; o->guidance_scale = 7.5f; o->steps = 25; o->batch_size = 1;  // impl_txt2img.rs:50-65
    o->callback_frequency = 1;
}


sde_status sde_txt2img_run(sde_engine* e, sde_scheduler* sch, const sde_txt2img_opts* o, float* images_out, sde_run_stats* stats)
{
    return guarded([&] {
        SDE_REQUIRE(e && sch && o, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(o->height % 8 == 0 && o->width % 8 == 0 && o->height > 0 && o->width > 0, SDE_ERR_INVALID_ARGUMENT,
                    "`width` (" + std::to_string(o->width) + ") and `height` (" + std::to_string(o->height) +
                        ") must be divisible by 8 for Stable Diffusion");  // impl_txt2img.rs:251-253
        SDE_REQUIRE(e->has_unet, SDE_ERR_MODEL, "engine was created without a UNet");
        SDE_REQUIRE(o->skip_decode || e->has_vae, SDE_ERR_MODEL, "engine was created without a VAE decoder");
        SDE_REQUIRE(o->skip_decode || images_out, SDE_ERR_INVALID_ARGUMENT, "images_out is null");
        SDE_REQUIRE(o->batch_size > 0 && o->steps > 0 && o->ehs && o->S > 0, SDE_ERR_INVALID_ARGUMENT, "bad batch/steps/embeddings");
        SDE_REQUIRE(o->D == e->unet.ctx_dim, SDE_ERR_INVALID_ARGUMENT, "text embedding width does not match the UNet's cross_attention_dim");
        EngineLock lk(e);
        engine_activate(e);
        const int B = o->batch_size, h = o->height / 8, w = o->width / 8, S = o->S, D = o->D, C = e->unet.in_channels;
        const bool cfg = o->guidance_scale > 1.0f;  // :258
        const int N = cfg ? 2 * B : B;
        // CFG-pair split: this engine runs one half of the CFG batch; the peer's half arrives through the exchange hook
        const int half = o->cfg_half;
        SDE_REQUIRE(half >= 0 && half <= 2, SDE_ERR_INVALID_ARGUMENT, "cfg_half must be 0, 1 or 2");
        SDE_REQUIRE(!half || (cfg && (o->exchange || o->exchange_stream)), SDE_ERR_INVALID_ARGUMENT, "cfg_half needs guidance_scale > 1 and an exchange hook");
        const int Nu = half ? B : N;                 // UNet batch on this engine
        const size_t half_off = half == 2 ? 1 : 0;   // which half of [2B] this engine owns
        const size_t per = (size_t)B * C * h * w;
        const int Cp = UNetModel::kInPad;
        hipStream_t st = e->stream;

        std::random_device rd;   // thread_rng().gen::<u64>() (:248): 64 fresh bits when no seed is given
        const uint64_t seed = o->has_seed ? o->seed : ((uint64_t)rd() << 32 | rd());
        // :249,267 -- StdRng::seed_from_u64(seed) / (seed + 31337), restated in rng.h (u64 add wraps in release builds)
        sde::ChaChaRng rng = sde::ChaChaRng::seed_from_u64(seed), sched_rng = sde::ChaChaRng::seed_from_u64(seed + 31337);

        sch->set_timesteps(o->steps);                         // :264
        const std::vector<float> ts = sch->timesteps;         // :269
        const int nt = (int)ts.size();
        // An empty schedule (the reference's DDIM with one inference step: ddim.rs:184 yields n - 1 timesteps).  The reference runs no
        // step and decodes init_noise_sigma * latents (impl_txt2img.rs:265,272,326); in parity mode (reference_quirks, the default)
        // so does the engine.  With the quirks off an empty schedule is a caller error.
        const bool empty = nt == 0;
        SDE_REQUIRE(!empty || sch->cfg.reference_quirks, SDE_ERR_SCHEDULER, "scheduler produced no timesteps");
        // per-timestep caller buffers: nt is the scheduler's count, not `steps` (DDPM: ceil(T / (T / steps)), ddpm.rs:181)
        if (o->noise_pred_out || o->latents_out) {
            const int rows = o->trace_rows > 0 ? o->trace_rows : o->steps;
            SDE_REQUIRE(nt <= rows, SDE_ERR_INVALID_ARGUMENT,
                        "noise_pred_out / latents_out hold " + std::to_string(rows) + " rows but this scheduler runs " + std::to_string(nt) +
                            " timesteps (set trace_rows from sde_scheduler_num_timesteps)");
        }
        if (o->step_noise) {
            const int rows = o->step_noise_rows > 0 ? o->step_noise_rows : o->steps;
            SDE_REQUIRE(nt <= rows, SDE_ERR_INVALID_ARGUMENT,
                        "step_noise holds " + std::to_string(rows) + " rows but this scheduler runs " + std::to_string(nt) +
                            " timesteps (set step_noise_rows from sde_scheduler_num_timesteps)");
        }
        const int i0 = o->start_step;   // image-to-image: the loop starts inside the schedule (sdengine.h)
        SDE_REQUIRE(i0 >= 0 && (i0 < nt || (empty && i0 == 0)), SDE_ERR_INVALID_ARGUMENT, "start_step must lie inside the scheduler's timesteps");
        SDE_REQUIRE(i0 == 0 || o->init_latents, SDE_ERR_INVALID_ARGUMENT, "start_step > 0 needs init_latents (the noised image latents)");
        long long num_warmup = (long long)nt - (long long)o->steps * 1;  // :270 ; negative == usize wrap-around in release builds
        if (num_warmup < 0) num_warmup = (1LL << 62);

        // ---- device state
        e->latents.ensure(per * 4);
        e->model_out.ensure((size_t)N * C * h * w * 4);
        e->x_packed.ensure((size_t)N * h * w * Cp * e->ctx.esize());
        HIP_CHECK(hipMemsetAsync(e->x_packed.p, 0, (size_t)N * h * w * Cp * e->ctx.esize(), st));
        e->ehs_h.ensure((size_t)N * S * D * e->ctx.esize());
        setup_kv(e, Nu, S);
        float* d_lat = e->latents.as<float>();
        std::vector<float> host_tmp;
        if (o->buffers_on_device) {
            launch_f32_to_f16(e->ctx, o->ehs, e->ehs_h.as<half_t>(), (int64_t)N * S * D);
        } else {
            e->ehs_f.ensure((size_t)N * S * D * 4);
            HIP_CHECK(hipMemcpyAsync(e->ehs_f.p, o->ehs, (size_t)N * S * D * 4, hipMemcpyHostToDevice, st));
            launch_f32_to_f16(e->ctx, e->ehs_f.as<float>(), e->ehs_h.as<half_t>(), (int64_t)N * S * D);
        }
        if (o->init_latents) {
            HIP_CHECK(hipMemcpyAsync(d_lat, o->init_latents, per * 4, o->buffers_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        } else {
            host_tmp.resize(per);
            sde::fill_standard_normal(rng, host_tmp.data(), per);           // :261-262
            HIP_CHECK(hipMemcpyAsync(d_lat, host_tmp.data(), per * 4, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st));
        }
        if (i0 == 0) launch_vec_op(e->ctx, d_lat, nullptr, d_lat, (int64_t)per, 2, sch->init_noise_sigma, 0.f);  // :265
        for (auto& hb : e->hist) hb.ensure(per * 4, true, st);
        float* hist[4] = {e->hist[0].as<float>(), e->hist[1].as<float>(), e->hist[2].as<float>(), e->hist[3].as<float>()};
        const bool want_taps = o->noise_pred_out != nullptr;
        if (want_taps) e->taps.ensure(per * 4);
        if (o->has_rescale_cfg && cfg) e->rescale_ws.ensure(4096);   // its own buffer: scratch_f belongs to sde_unet_forward
        // Scheduler noise (Euler-a, DDPM): draw t's tensor on the host while the GPU is inside UNet call t, into a slot of its own in
        // page-locked memory, and copy it with the stream -- no per-step synchronisation (all nt x per draws come from one
        // sequential stream, seed + 31337, so they are drawn in step order).  Very long runs fall back to a two-slot ring.
        const bool host_noise = !(o->step_noise && o->buffers_on_device);
        const size_t noise_slots = (size_t)nt * per * 4 <= ((size_t)512 << 20) ? (size_t)nt : 2;
        hipEvent_t noise_ev[2] = {nullptr, nullptr};
        struct NoiseEvents { hipEvent_t* e; ~NoiseEvents() { for (int k = 0; k < 2; ++k) if (e[k]) (void)hipEventDestroy(e[k]); } } noise_evs{noise_ev};

        struct Events {   // destroyed on every exit path (an error inside the loop throws past the end of this function)
            hipEvent_t e[3] = {nullptr, nullptr, nullptr};
            ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
        } evs;
        for (hipEvent_t& x : evs.e) HIP_CHECK(hipEventCreate(&x));
        const hipEvent_t ev0 = evs.e[0], ev1 = evs.e[1], ev2 = evs.e[2];
        HIP_CHECK(hipEventRecord(ev0, st));

        // first model input: scale_model_input(concat(latents, latents), t0)  (:273-278)
        if (!empty) {
            float div = sch->input_scale_divisor(ts[(size_t)i0]);
            launch_pack_latents(e->ctx, d_lat, e->x_packed.as<half_t>(), B, (cfg && !half) ? 2 : 1, C, h, w, Cp, 1.0f / div);
        }
        int unet_calls = 0, graph_launches = 0;
        double enqueue_ms = 0.0;
        struct LaneDrain {   // nothing of the preview lane may outlive the call, whichever way it ends
            sde_engine* e;
            ~LaneDrain() { if (e->preview) (void)hipStreamSynchronize(e->preview->stream); }
        } lane_drain{e};
        bool pending_cb = false, stopped = false;   // a payload callback produced at step pending_i, not yet delivered
        int pending_i = 0; float pending_t = 0.f; size_t pending_n = 0;
        for (int i = i0; i < nt; ++i) {
            const float t = ts[(size_t)i];
            const auto tq0 = std::chrono::steady_clock::now();
            graph_launches += unet_run(e, e->x_packed.as<half_t>(), Nu, h, w, t, (const half_t*)(e->ehs_h.as<char>() + half_off * B * S * D * e->ctx.esize()), S,
                                       i == i0, e->model_out.as<float>() + half_off * per) ? 1 : 0;                                             // :283
            enqueue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count();
            if (pending_cb) {
                // deliver step i-1's payload while the GPU works on UNet call i.  `false` -> break (:319-321): the UNet call just
                // enqueued is abandoned (it only wrote model_out), latents are still those of step i-1, as in the reference.
                HIP_CHECK(hipEventSynchronize(e->preview->done));
                pending_cb = false;
                if (!o->callback(pending_i, pending_t, e->preview->host.as<float>(), pending_n, o->callback_user)) { stopped = true; break; }
            }
            ++unet_calls;
            if (half && o->exchange_stream) {
                // stream-ordered: the hook enqueues the peer transfer on the engine's stream (or makes it wait); nothing drains
                SDE_REQUIRE(o->exchange_stream(e->model_out.as<float>(), per, (int)half_off, (void*)st, o->exchange_user) != 0, SDE_ERR_EXCHANGE,
                            "CFG-pair exchange hook failed at step " + std::to_string(i));
            } else if (half) {
                HIP_CHECK(hipStreamSynchronize(st));
                SDE_REQUIRE(o->exchange(e->model_out.as<float>(), per, (int)half_off, o->exchange_user) != 0, SDE_ERR_EXCHANGE,
                            "CFG-pair exchange hook failed at step " + std::to_string(i));
            }
            StepPlan p;
            sch->plan(t, p);
            StepArgs a;
            a.kind = p.kind; a.pred = p.pred; a.B = B; a.C = C; a.H = h; a.W = w; a.cfg = cfg ? 1 : 0; a.guidance = o->guidance_scale;
            a.model_out = e->model_out.as<float>(); a.latents = d_lat;
            for (int k = 0; k < 12; ++k) a.c[k] = p.c[k];
            a.clip = p.clip; a.nhist = p.nhist; a.dpm_algo = p.dpm_algo;
            for (int k = 0; k < 4; ++k) { a.lms_c[k] = p.lms_c[k]; a.hist[k] = hist[k]; }
            if (p.needs_noise) {
                if (!host_noise) a.noise = o->step_noise + (size_t)i * per;
                else {
                    e->noise.ensure(noise_slots * per * 4);
                    e->noise_host.ensure(noise_slots * per * 4);
                    const size_t slot = noise_slots == (size_t)nt ? (size_t)i : (size_t)(i & 1);
                    float* hbuf = e->noise_host.as<float>() + slot * per;
                    if (noise_slots == 2) {   // ring: the copy that last read this slot must have finished
                        if (noise_ev[slot]) HIP_CHECK(hipEventSynchronize(noise_ev[slot]));
                        else HIP_CHECK(hipEventCreateWithFlags(&noise_ev[slot], hipEventDisableTiming));
                    }
                    if (o->step_noise) memcpy(hbuf, o->step_noise + (size_t)i * per, per * 4);
                    else sde::fill_standard_normal(sched_rng, hbuf, per);
                    float* dbuf = e->noise.as<float>() + slot * per;
                    HIP_CHECK(hipMemcpyAsync(dbuf, hbuf, per * 4, hipMemcpyHostToDevice, st));
                    if (noise_slots == 2) HIP_CHECK(hipEventRecord(noise_ev[slot], st));
                    a.noise = dbuf;
                }
            }
            if (o->has_rescale_cfg && cfg) {
                float* ws = e->rescale_ws.as<float>();
                launch_cfg_rescale_factor(e->ctx, e->model_out.as<float>(), B, (int64_t)per, o->guidance_scale, o->rescale_cfg, ws + 16, ws);
                a.rescale_factor = ws;
            }
            if (i + 1 < nt) {
                a.next_input = e->x_packed.as<half_t>(); a.Cpad = Cp; a.next_input_f32 = e->ctx.f32 ? 1 : 0;
                a.next_input_scale = 1.0f / sch->input_scale_divisor(ts[(size_t)i + 1]);
            }
            if (want_taps) a.eps_out = e->taps.as<float>();
            launch_cfg_step(e->ctx, a);                                                                  // :288-304
            sch->advance(p);
            sch->rotate_history(p, hist);
            if (want_taps) HIP_CHECK(hipMemcpyAsync(o->noise_pred_out + (size_t)i * per, e->taps.p, per * 4, hipMemcpyDeviceToHost, st));
            if (o->latents_out) HIP_CHECK(hipMemcpyAsync(o->latents_out + (size_t)i * per, d_lat, per * 4, hipMemcpyDeviceToHost, st));
            if (want_taps || o->latents_out) HIP_CHECK(hipStreamSynchronize(st));
            // callbacks (:306-323).  Progress needs nothing from the GPU and is called at once.  The kinds that hand over a tensor
            // are produced on the preview lane (second stream) from the latents this step left, and DELIVERED after the next UNet
            // call has been enqueued, so the denoise stream never waits for the copy, the VAE or the user's code; the callback
            // still sees step i's latents, in step order, on the caller's thread, before step i + 1's scheduler update.
            if (o->callback && o->callback_kind != SDE_CB_NONE) {
                const int freq = std::max(1, o->callback_frequency);
                if (i == nt - 1 || ((long long)(i + 1) > num_warmup && (i + 1) % 1 == 0)) {
                    const int kind = o->callback_kind;
                    if (kind == SDE_CB_PROGRESS && i % freq == 0) {
                        if (!o->callback(i, t, nullptr, 0, o->callback_user)) { stopped = true; break; }  // :319-321, still decodes
                    } else if ((kind == SDE_CB_LATENTS && i % freq == 0) ||
                               ((kind == SDE_CB_DECODED || kind == SDE_CB_APPROXIMATE) && i != 0 && i % freq == 0)) {
                        PreviewLane& L = preview_lane(e);
                        HIP_CHECK(hipEventRecord(L.ready, st));
                        HIP_CHECK(hipStreamWaitEvent(L.stream, L.ready, 0));
                        size_t no = per;
                        const float* src = d_lat;
                        if (kind == SDE_CB_DECODED) {
                            const int up = e->vae.upscale();
                            no = (size_t)B * h * up * w * up * e->vae.out_channels;
                            L.images.ensure(no * 4);
                            decode_on_lane(e, L, d_lat, B, h, w, L.images.as<float>());
                            src = L.images.as<float>();
                        } else if (kind == SDE_CB_APPROXIMATE) {
                            no = (size_t)B * h * w * 3;
                            L.images.ensure(no * 4);
                            launch_approx_decode(L.ctx, d_lat, L.images.as<float>(), B, h, w);
                            src = L.images.as<float>();
                        }
                        L.host.ensure(no * 4);
                        HIP_CHECK(hipMemcpyAsync(L.host.p, src, no * 4, hipMemcpyDeviceToHost, L.stream));
                        HIP_CHECK(hipEventRecord(L.done, L.stream));
                        pending_cb = true; pending_i = i; pending_t = t; pending_n = no;
                    }
                }
            }
        }
        HIP_CHECK(hipEventRecord(ev1, st));
        if (o->final_latents_out) HIP_CHECK(hipMemcpyAsync(o->final_latents_out, d_lat, per * 4, hipMemcpyDeviceToHost, st));
        if (!o->skip_decode) {                                                                           // :326
            const int up = e->vae.upscale();  // 8 for Stable Diffusion VAEs => [B,H,W,3]
            const size_t img = (size_t)h * up * w * up * e->vae.out_channels;
            // CFG-pair split: both engines hold the same latents; each decodes its share of the batch (rows [b0, b0+nb))
            const int B0 = (B + 1) / 2;
            const int b0 = half == 2 ? B0 : 0, nb = half == 0 ? B : (half == 1 ? B0 : B - B0);
            const size_t no = (size_t)nb * img;
            if (nb > 0) {
                if (o->buffers_on_device) decode_dev(e, d_lat + (size_t)b0 * C * h * w, nb, h, w, EPI_IMAGE, images_out + (size_t)b0 * img);
                else {
                    e->images.ensure(no * 4);
                    decode_dev(e, d_lat + (size_t)b0 * C * h * w, nb, h, w, EPI_IMAGE, e->images.as<float>());
                }
            }
            HIP_CHECK(hipEventRecord(ev2, st));
            if (!o->buffers_on_device && nb > 0)
                HIP_CHECK(hipMemcpyAsync(images_out + (size_t)b0 * img, e->images.p, no * 4, hipMemcpyDeviceToHost, st));
        } else HIP_CHECK(hipEventRecord(ev2, st));
        if (pending_cb && !stopped) {   // the last step's payload (i == nt - 1 always fires, :307), delivered while the final decode runs
            HIP_CHECK(hipEventSynchronize(e->preview->done));
            pending_cb = false;
            (void)o->callback(pending_i, pending_t, e->preview->host.as<float>(), pending_n, o->callback_user);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        if (stats) {
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, ev0, ev1); (void)hipEventElapsedTime(&b, ev1, ev2);
            stats->unet_calls = unet_calls; stats->unet_ms = a; stats->vae_ms = b; stats->total_ms = a + b;
            stats->host_enqueue_ms = (float)enqueue_ms;
            stats->unet_graph_launches = graph_launches;
        }
    });
}

// ------------------------------------------------------------------------------------------ C ABI: measurement + test hooks
sde_status sde_profile_enable(sde_engine* e, int enable)
{
    return guarded([&] { SDE_REQUIRE(e, SDE_ERR_INVALID_ARGUMENT, "null engine"); engine_activate(e); e->prof.resolve(); e->prof.enabled = enable != 0; });
}
sde_status sde_profile_reset(sde_engine* e)
{
    return guarded([&] { SDE_REQUIRE(e, SDE_ERR_INVALID_ARGUMENT, "null engine"); engine_activate(e); e->prof.resolve(); e->prof.classes.clear(); });
}
sde_status sde_profile_get(sde_engine* e, sde_kernel_stat* out, int capacity, int* n_out)
{
    return guarded([&] {
        SDE_REQUIRE(e && n_out, SDE_ERR_INVALID_ARGUMENT, "null argument");
        engine_activate(e);
        e->prof.resolve();
        *n_out = (int)e->prof.classes.size();
        for (int i = 0; out && i < capacity && i < *n_out; ++i) {
            const auto& c = e->prof.classes[(size_t)i];
            memset(&out[i], 0, sizeof(out[i]));
            strncpy(out[i].name, c.name.c_str(), sizeof(out[i].name) - 1);
            out[i].launches = c.launches; out[i].total_ms = c.ms; out[i].flops = c.flops; out[i].bytes = c.bytes;
        }
    });
}
sde_status sde_synchronize(sde_engine* e)
{
    return guarded([&] { SDE_REQUIRE(e, SDE_ERR_INVALID_ARGUMENT, "null engine"); engine_activate(e); HIP_CHECK(hipStreamSynchronize(e->stream)); });
}

// The test hooks compute their output shape themselves and copy that many floats into a caller buffer: the caller states the
// buffer's capacity (elements) and anything else than an exact fit is refused before the device-to-host copy.
static void require_out_capacity(const char* fn, size_t need, size_t have, const std::string& shape)
{
    SDE_REQUIRE(need == have, SDE_ERR_INVALID_ARGUMENT,
                std::string(fn) + ": the output is " + shape + " = " + std::to_string(need) + " floats, the caller's buffer holds " +
                    std::to_string(have) + " (out_capacity)");
}

sde_status sde_test_conv(sde_engine* e, int naive, const float* x, int NB, int Hi, int Wi, int Cin, const float* w, const float* bias, int Cout,
                         int ksize, int stride, int upsample, float* out, size_t out_capacity)
{
    return guarded([&] {
        SDE_REQUIRE(e && x && w && out, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(ksize == 1 || ksize == 3, SDE_ERR_INVALID_ARGUMENT, "ksize must be 1 or 3");
        SDE_REQUIRE(NB > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0 && stride >= 1 && stride <= 2, SDE_ERR_INVALID_ARGUMENT, "bad conv geometry");
        EngineLock lk(e);
        engine_activate(e);
        LaunchCtx ctx = e->ctx; ctx.f32 = false; ctx.force_naive = naive == 1; ctx.force_big = naive == 2; ctx.force_mid = naive == 3; ctx.force_mid128 = naive == 4; ctx.force_midh = naive == 5; ctx.force_ws = naive == 6;   // 0 fast, 1 plain, 2 / 3 / 4 / 5 / 6 force the 256x320 / 256x160 / 128x160 / halo / W-stationary kernel
        const int taps = ksize * ksize;
        // upsample: 1 = nearest 2x; 2 = 2x cropped to (2*Hi - 1) x (2*Wi - 1), the UNet's upsample-to-skip-size on odd latents
        // upsample == 3: not an upsample -- the VAE encoder's downsampler, pad (0,1,0,1) + a pad-0 stride-2 window (GemmArgs::coff = 1)
        const bool asym = upsample == 3;
        SDE_REQUIRE(!asym || (ksize == 3 && stride == 2 && Hi >= 2 && Wi >= 2), SDE_ERR_INVALID_ARGUMENT, "asymmetric padding goes with a 3x3 stride-2 conv");
        if (asym) upsample = 0;
        const int Hv = upsample ? 2 * Hi - (upsample == 2) : Hi, Wv = upsample ? 2 * Wi - (upsample == 2) : Wi;
        const int Ho = asym ? (Hi - 2) / 2 + 1 : (ksize == 3 ? (Hv - 1) / stride + 1 : Hv), Wo = asym ? (Wi - 2) / 2 + 1 : (ksize == 3 ? (Wv - 1) / stride + 1 : Wv);
        const size_t nx = (size_t)NB * Hi * Wi * Cin, nw = (size_t)Cout * Cin * taps, no = (size_t)NB * Ho * Wo * Cout;
        // the output geometry is computed HERE; the caller states what its buffer holds and a disagreement is refused before any copy
        require_out_capacity("sde_test_conv", no, out_capacity,
                             "[" + std::to_string(NB) + "," + std::to_string(Ho) + "," + std::to_string(Wo) + "," + std::to_string(Cout) + "]");
        const int Npad = gemm_npad(Cout), K = taps * Cin;
        // naive 7 / 8: the fp32 engine's kernels (f32path.hip), MFMA tiles / plain FMA: no fp16 anywhere, operands stay floats
        const bool f32k = naive == 7 || naive == 8;
        ctx.f32 = f32k; if (naive == 8) ctx.force_naive = true;
        const size_t es = f32k ? 4 : 2;
        DevBuf dxf, dxh, dwf, dwh, dbias, doh, dof;
        dxf.ensure(nx * 4); dxh.ensure(nx * 2); dwf.ensure(nw * 4); dwh.ensure((size_t)Npad * K * es); doh.ensure(no * 2); dof.ensure(no * 4);
        HIP_CHECK(hipMemcpy(dxf.p, x, nx * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dwf.p, w, nw * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemsetAsync(dwh.p, 0, (size_t)Npad * K * es, e->stream));
        if (bias) { dbias.ensure((size_t)Cout * 4); HIP_CHECK(hipMemcpy(dbias.p, bias, (size_t)Cout * 4, hipMemcpyHostToDevice)); }
        if (!f32k) launch_f32_to_f16(ctx, dxf.as<float>(), dxh.as<half_t>(), (int64_t)nx);
        RepackArgs r;
        r.src = dwf.p; r.src_is_f16 = 0; r.O = Cout; r.I = Cin; r.KH = ksize; r.KW = ksize; r.in_out = 0; r.dst = dwh.as<half_t>();
        r.dst_row0 = 0; r.Ipad = Cin; r.i0 = 0; r.Kdst = K; r.dst_f32 = f32k ? 1 : 0;
        launch_repack(ctx, r);
        GemmArgs g;
        g.A0 = f32k ? dxf.as<half_t>() : dxh.as<half_t>(); g.lda0 = Cin; g.C0 = Cin; g.taps = taps; g.stride = ksize == 3 ? stride : 1; g.ups = ksize == 3 ? (upsample != 0) : 0;
        g.coff = asym ? 1 : 0;
        g.Hi = Hv; g.Wi = Wv; g.Ho = Ho; g.Wo = Wo; g.M = NB * Ho * Wo; g.N = Cout; g.K = K;
        g.W = dwh.as<half_t>(); g.Npad = Npad; g.bias = bias ? dbias.as<float>() : nullptr; g.D = f32k ? dof.as<half_t>() : doh.as<half_t>(); g.ldd = Cout;
        launch_gemm(ctx, g);
        if (!f32k) launch_f16_to_f32(ctx, doh.as<half_t>(), dof.as<float>(), (int64_t)no);
        HIP_CHECK(hipStreamSynchronize(e->stream));
        HIP_CHECK(hipMemcpy(out, dof.p, no * 4, hipMemcpyDeviceToHost));
    });
}

sde_status sde_test_conv_res(sde_engine* e, int naive, const float* x, int NB, int Hi, int Wi, int Cin, const float* w, const float* bias, int Cout,
                             int ksize, const float* residual, int split, float* out, size_t out_capacity)
{
    return guarded([&] {
        SDE_REQUIRE(e && x && w && out && residual, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(ksize == 1 || ksize == 3, SDE_ERR_INVALID_ARGUMENT, "ksize must be 1 or 3");
        SDE_REQUIRE(NB > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, SDE_ERR_INVALID_ARGUMENT, "bad conv geometry");
        require_out_capacity("sde_test_conv_res", (size_t)NB * Hi * Wi * Cout, out_capacity,
                             "[" + std::to_string(NB) + "," + std::to_string(Hi) + "," + std::to_string(Wi) + "," + std::to_string(Cout) + "]");
        EngineLock lk(e);
        engine_activate(e);
        LaunchCtx ctx = e->ctx; ctx.f32 = false;
        ctx.force_naive = naive == 1; ctx.force_big = naive == 2; ctx.force_mid = naive == 3; ctx.force_mid128 = naive == 4; ctx.force_midh = naive == 5;
        ctx.force_ws = naive == 6;
        const int taps = ksize * ksize;
        const size_t nx = (size_t)NB * Hi * Wi * Cin, nw = (size_t)Cout * Cin * taps, no = (size_t)NB * Hi * Wi * Cout;
        const int Npad = gemm_npad(Cout), K = taps * Cin;
        // the residual as fp16 hi (+ lo): split on the host, the way a producer's epilogue would have left it
        std::vector<half_t> rhi(no), rlo(no);
        for (size_t i = 0; i < no; ++i) { rhi[i] = (half_t)residual[i]; rlo[i] = (half_t)(residual[i] - (float)rhi[i]); }
        DevBuf dxf, dxh, dwf, dwh, dbias, doh, dol, drh, drl;
        dxf.ensure(nx * 4); dxh.ensure(nx * 2); dwf.ensure(nw * 4); dwh.ensure((size_t)Npad * K * 2); doh.ensure(no * 2); dol.ensure(no * 2);
        drh.ensure(no * 2); drl.ensure(no * 2);
        HIP_CHECK(hipMemcpy(dxf.p, x, nx * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dwf.p, w, nw * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(drh.p, rhi.data(), no * 2, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(drl.p, rlo.data(), no * 2, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemsetAsync(dwh.p, 0, (size_t)Npad * K * 2, e->stream));
        HIP_CHECK(hipMemsetAsync(dol.p, 0, no * 2, e->stream));
        if (bias) { dbias.ensure((size_t)Cout * 4); HIP_CHECK(hipMemcpy(dbias.p, bias, (size_t)Cout * 4, hipMemcpyHostToDevice)); }
        launch_f32_to_f16(ctx, dxf.as<float>(), dxh.as<half_t>(), (int64_t)nx);
        RepackArgs r;
        r.src = dwf.p; r.src_is_f16 = 0; r.O = Cout; r.I = Cin; r.KH = ksize; r.KW = ksize; r.in_out = 0; r.dst = dwh.as<half_t>();
        r.dst_row0 = 0; r.Ipad = Cin; r.i0 = 0; r.Kdst = K;
        launch_repack(ctx, r);
        GemmArgs g;
        g.A0 = dxh.as<half_t>(); g.lda0 = Cin; g.C0 = Cin; g.taps = taps; g.stride = 1;
        g.Hi = Hi; g.Wi = Wi; g.Ho = Hi; g.Wo = Wi; g.M = NB * Hi * Wi; g.N = Cout; g.K = K;
        g.W = dwh.as<half_t>(); g.Npad = Npad; g.bias = bias ? dbias.as<float>() : nullptr; g.D = doh.as<half_t>(); g.ldd = Cout;
        g.R = drh.as<half_t>(); g.ldr = Cout;
        if (split) { g.Rlo = drl.as<half_t>(); g.Dlo = dol.as<half_t>(); }
        launch_gemm(ctx, g);
        HIP_CHECK(hipStreamSynchronize(e->stream));
        HIP_CHECK(hipMemcpy(rhi.data(), doh.p, no * 2, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(rlo.data(), dol.p, no * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < no; ++i) out[i] = (float)rhi[i] + (split ? (float)rlo[i] : 0.f);
    });
}

sde_status sde_test_geglu(sde_engine* e, int naive, const float* x, int M, int K, const float* w, const float* bias, int N, int gelu_tanh,
                          float* out, size_t out_capacity)
{
    return guarded([&] {
        SDE_REQUIRE(e && x && w && out, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(M > 0 && K > 0 && N > 0 && (N / 2) % 16 == 0 && N % 2 == 0, SDE_ERR_INVALID_ARGUMENT, "GEGLU projection width must be a multiple of 32");
        require_out_capacity("sde_test_geglu", (size_t)M * (N / 2), out_capacity, "[" + std::to_string(M) + "," + std::to_string(N / 2) + "]");
        EngineLock lk(e);
        engine_activate(e);
        LaunchCtx ctx = e->ctx; ctx.f32 = false;
        ctx.force_naive = naive == 1; ctx.force_big = naive == 2; ctx.force_ws = naive == 6;
        const bool pair = N % 320 == 0 && K % 64 == 0;      // the loader's rule (WeightBuilder::linear)
        const size_t nx = (size_t)M * K, nw = (size_t)N * K, no = (size_t)M * (N / 2);
        const int Npad = gemm_npad(N);
        DevBuf dxf, dxh, dwf, dwh, dbias, doh, dof;
        dxf.ensure(nx * 4); dxh.ensure(nx * 2); dwf.ensure(nw * 4); dwh.ensure((size_t)Npad * K * 2); doh.ensure(no * 2); dof.ensure(no * 4);
        HIP_CHECK(hipMemcpy(dxf.p, x, nx * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dwf.p, w, nw * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemsetAsync(dwh.p, 0, (size_t)Npad * K * 2, e->stream));
        if (bias) { dbias.ensure((size_t)N * 4); HIP_CHECK(hipMemcpy(dbias.p, bias, (size_t)N * 4, hipMemcpyHostToDevice)); }
        launch_f32_to_f16(ctx, dxf.as<float>(), dxh.as<half_t>(), (int64_t)nx);
        RepackArgs r;
        r.src = dwf.p; r.src_is_f16 = 0; r.O = N; r.I = K; r.KH = 1; r.KW = 1; r.in_out = 0; r.dst = dwh.as<half_t>();
        r.dst_row0 = 0; r.Ipad = K; r.i0 = 0; r.Kdst = K; r.geglu_half = N / 2; r.geglu_pair = pair ? 1 : 0;
        launch_repack(ctx, r);
        GemmArgs g;
        g.A0 = dxh.as<half_t>(); g.lda0 = K; g.C0 = K; g.taps = 1; g.M = M; g.N = N; g.K = K;
        g.W = dwh.as<half_t>(); g.Npad = Npad; g.bias = bias ? dbias.as<float>() : nullptr; g.D = doh.as<half_t>(); g.ldd = N / 2;
        g.epi = pair ? EPI_GEGLU_PAIR : EPI_GEGLU; g.gelu_tanh = gelu_tanh ? 1 : 0;
        launch_gemm(ctx, g);
        launch_f16_to_f32(ctx, doh.as<half_t>(), dof.as<float>(), (int64_t)no);
        HIP_CHECK(hipStreamSynchronize(e->stream));
        HIP_CHECK(hipMemcpy(out, dof.p, no * 4, hipMemcpyDeviceToHost));
    });
}

sde_status sde_test_attention(sde_engine* e, int naive, const float* q, const float* k, const float* v, int NB, int heads, int Sq, int Skv, int d,
                              float* out, size_t out_capacity)
{
    return guarded([&] {
        SDE_REQUIRE(e && q && k && v && out, SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(NB > 0 && heads > 0 && Sq > 0 && Skv > 0 && d > 0, SDE_ERR_INVALID_ARGUMENT, "bad attention geometry");
        require_out_capacity("sde_test_attention", (size_t)NB * Sq * heads * d, out_capacity,
                             "[" + std::to_string(NB) + "," + std::to_string(Sq) + "," + std::to_string(heads * d) + "]");
        EngineLock lk(e);
        engine_activate(e);
        const bool f32k = naive == 7 || naive == 8;          // the fp32 engine's kernels: 7 = MFMA, 8 = plain
        LaunchCtx ctx = e->ctx; ctx.f32 = f32k; ctx.force_naive = f32k ? naive == 8 : naive != 0;
        const int C = heads * d;
        const size_t nq = (size_t)NB * Sq * C, nk = (size_t)NB * Skv * C;
        DevBuf f, qh, kh, vh, oh, of;
        f.ensure(std::max(nq, nk) * 4); qh.ensure(nq * 2); kh.ensure(nk * 2); vh.ensure(nk * 2); oh.ensure(nq * 2); of.ensure(nq * 4);
        auto up = [&](const float* src, size_t n, DevBuf& dst) {
            HIP_CHECK(hipMemcpy(f.p, src, n * 4, hipMemcpyHostToDevice));
            launch_f32_to_f16(ctx, f.as<float>(), dst.as<half_t>(), (int64_t)n);
            HIP_CHECK(hipStreamSynchronize(e->stream));
        };
        if (f32k) {
            qh.ensure(nq * 4); kh.ensure(nk * 4); vh.ensure(nk * 4);
            HIP_CHECK(hipMemcpy(qh.p, q, nq * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(kh.p, k, nk * 4, hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(vh.p, v, nk * 4, hipMemcpyHostToDevice));
        } else { up(q, nq, qh); up(k, nk, kh); up(v, nk, vh); }
        AttnArgs a;
        a.Q = qh.as<half_t>(); a.K = kh.as<half_t>(); a.V = vh.as<half_t>(); a.O = f32k ? of.as<half_t>() : oh.as<half_t>();
        a.ldq = a.ldk = a.ldv = a.ldo = C; a.NB = NB; a.heads = heads; a.Sq = Sq; a.Skv = Skv; a.d = d; a.scale = 1.0f / sqrtf((float)d);
        launch_attention(ctx, a);
        if (!f32k) launch_f16_to_f32(ctx, oh.as<half_t>(), of.as<float>(), (int64_t)nq);
        HIP_CHECK(hipStreamSynchronize(e->stream));
        HIP_CHECK(hipMemcpy(out, of.p, nq * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
