// loader.cpp — see loader.h
#include "loader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <fstream>
#include <sstream>

// ------------------------------------------------------------------------------------------ mmap
MappedFile::MappedFile(const std::string& path)
{
    fd = ::open(path.c_str(), O_RDONLY);
    SDE_REQUIRE(fd >= 0, SDE_ERR_IO, "cannot open " + path + ": " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); sde_fail(SDE_ERR_IO, "cannot stat " + path); }
    size = (size_t)st.st_size;
    if (size == 0) { ::close(fd); sde_fail(SDE_ERR_IO, path + " is empty"); }
    void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) { ::close(fd); sde_fail(SDE_ERR_IO, "cannot mmap " + path); }
    data = (const uint8_t*)p;
}
MappedFile::~MappedFile()
{
    if (data) munmap((void*)data, size);
    if (fd >= 0) ::close(fd);
}

// ------------------------------------------------------------------------------------------ protobuf wire reader
namespace {
struct Reader {
    const uint8_t* p; const uint8_t* end;
    bool done() const { return p >= end; }
    uint64_t varint()
    {
        uint64_t r = 0; int shift = 0;
        while (true) {
            SDE_REQUIRE(p < end && shift < 70, SDE_ERR_MODEL, "onnx: truncated varint");
            uint8_t b = *p++;
            r |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) break;
            shift += 7;
        }
        return r;
    }
    // returns field number, fills wire type; for wt 2 sets sub to the payload
    int next(int& wt, uint64_t& val, Reader& sub)
    {
        uint64_t key = varint();
        wt = (int)(key & 7);
        int fn = (int)(key >> 3);
        switch (wt) {
        case 0: val = varint(); break;
        case 1: SDE_REQUIRE(p + 8 <= end, SDE_ERR_MODEL, "onnx: truncated fixed64"); memcpy(&val, p, 8); p += 8; break;
        case 2: {
            uint64_t len = varint();
            SDE_REQUIRE(len <= (uint64_t)(end - p), SDE_ERR_MODEL, "onnx: truncated length-delimited field");
            sub.p = p; sub.end = p + len; p += len;
        } break;
        case 5: { SDE_REQUIRE(p + 4 <= end, SDE_ERR_MODEL, "onnx: truncated fixed32"); uint32_t v; memcpy(&v, p, 4); val = v; p += 4; } break;
        default: sde_fail(SDE_ERR_MODEL, "onnx: unsupported wire type " + std::to_string(wt));
        }
        return fn;
    }
};
std::string str_of(const Reader& r) { return std::string((const char*)r.p, (size_t)(r.end - r.p)); }

void packed_ints(int wt, uint64_t val, Reader sub, std::vector<int64_t>& out)
{
    if (wt == 0) { out.push_back((int64_t)val); return; }
    while (!sub.done()) out.push_back((int64_t)sub.varint());
}

struct ExternalInfo { std::string location; int64_t offset = 0, length = -1; bool present = false; };

void parse_tensor(Reader r, OnnxTensor& t, ExternalInfo& ext)
{
    std::vector<float> fdata; std::vector<int64_t> i32data, i64data;
    int wt; uint64_t v; Reader sub{};
    while (!r.done()) {
        int fn = r.next(wt, v, sub);
        switch (fn) {
        case 1: packed_ints(wt, v, sub, t.dims); break;
        case 2: t.data_type = (int)v; break;
        case 4:
            if (wt == 5) { float f; uint32_t u = (uint32_t)v; memcpy(&f, &u, 4); fdata.push_back(f); }
            else { size_t n = (size_t)(sub.end - sub.p) / 4; size_t o = fdata.size(); fdata.resize(o + n); memcpy(fdata.data() + o, sub.p, n * 4); }
            break;
        case 5: packed_ints(wt, v, sub, i32data); break;
        case 7: packed_ints(wt, v, sub, i64data); break;
        case 8: t.name = str_of(sub); break;
        case 9: t.data = sub.p; t.nbytes = (size_t)(sub.end - sub.p); break;
        case 13: {
            std::string k, val; int wt2; uint64_t v2; Reader s2{};
            while (!sub.done()) { int f2 = sub.next(wt2, v2, s2); if (f2 == 1) k = str_of(s2); else if (f2 == 2) val = str_of(s2); }
            ext.present = true;
            if (k == "location") ext.location = val;
            else if (k == "offset") ext.offset = std::stoll(val);
            else if (k == "length") ext.length = std::stoll(val);
        } break;
        case 14: if (v == 1) ext.present = true; break;
        default: break;
        }
    }
    if (!t.data && !ext.present) {
        if (!fdata.empty()) { t.owned.resize(fdata.size() * 4); memcpy(t.owned.data(), fdata.data(), t.owned.size()); }
        else if (!i64data.empty()) { t.owned.resize(i64data.size() * 8); memcpy(t.owned.data(), i64data.data(), t.owned.size()); }
        else if (!i32data.empty()) {
            if (t.data_type == 10) {  // fp16 bit patterns carried in int32_data (scripts/_optimizer/float16.py:93-108)
                t.owned.resize(i32data.size() * 2);
                for (size_t i = 0; i < i32data.size(); ++i) { uint16_t h = (uint16_t)i32data[i]; memcpy(t.owned.data() + 2 * i, &h, 2); }
            } else {
                t.owned.resize(i32data.size() * 4);
                for (size_t i = 0; i < i32data.size(); ++i) { int32_t x = (int32_t)i32data[i]; memcpy(t.owned.data() + 4 * i, &x, 4); }
            }
        }
        t.data = t.owned.data(); t.nbytes = t.owned.size();
    }
}

void parse_attr(Reader r, OnnxAttr& a)
{
    int wt; uint64_t v; Reader sub{};
    while (!r.done()) {
        int fn = r.next(wt, v, sub);
        switch (fn) {
        case 1: a.name = str_of(sub); break;
        case 2: { uint32_t u = (uint32_t)v; memcpy(&a.f, &u, 4); a.has_f = true; } break;
        case 3: a.i = (int64_t)v; a.has_i = true; break;
        case 4: a.s = str_of(sub); break;
        case 5: { a.t = std::make_shared<OnnxTensor>(); ExternalInfo e; parse_tensor(sub, *a.t, e); } break;
        case 8: packed_ints(wt, v, sub, a.ints); break;
        default: break;
        }
    }
}

void parse_node(Reader r, OnnxNode& n)
{
    int wt; uint64_t v; Reader sub{};
    while (!r.done()) {
        int fn = r.next(wt, v, sub);
        switch (fn) {
        case 1: n.inputs.push_back(str_of(sub)); break;
        case 2: n.outputs.push_back(str_of(sub)); break;
        case 3: n.name = str_of(sub); break;
        case 4: n.op_type = str_of(sub); break;
        case 5: n.attrs.emplace_back(); parse_attr(sub, n.attrs.back()); break;
        case 7: n.domain = str_of(sub); break;
        default: break;
        }
    }
}

size_t dtype_size(int dt)
{
    switch (dt) { case 1: return 4; case 10: return 2; case 16: return 2; case 7: return 8; case 11: return 8; case 6: return 4; case 9: return 1; case 2: return 1; case 3: return 1; default: return 0; }
}
std::string dirname_of(const std::string& p)
{
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}
}  // namespace

void onnx_load(const std::string& path, OnnxModel& m)
{
    m.files.emplace_back(new MappedFile(path));
    const MappedFile& mf = *m.files.back();
    const std::string base = dirname_of(path);
    std::map<std::string, const MappedFile*> ext_files;
    Reader r{mf.data, mf.data + mf.size};
    int wt; uint64_t v; Reader sub{};
    bool saw_graph = false;
    while (!r.done()) {
        int fn = r.next(wt, v, sub);
        if (fn == 1) m.ir_version = (int64_t)v;
        else if (fn == 2) m.producer = str_of(sub);
        else if (fn == 8) {
            std::string dom; int64_t ver = 0; int wt2; uint64_t v2; Reader s2{};
            while (!sub.done()) { int f2 = sub.next(wt2, v2, s2); if (f2 == 1) dom = str_of(s2); else if (f2 == 2) ver = (int64_t)v2; }
            m.opsets[dom] = ver;
        } else if (fn == 7) {
            saw_graph = true;
            Reader g = sub; int wt2; uint64_t v2; Reader s2{};
            while (!g.done()) {
                int f2 = g.next(wt2, v2, s2);
                if (f2 == 1) { m.nodes.emplace_back(); parse_node(s2, m.nodes.back()); }
                else if (f2 == 5) {
                    OnnxTensor t; ExternalInfo e;
                    parse_tensor(s2, t, e);
                    if (e.present) {
                        SDE_REQUIRE(!e.location.empty(), SDE_ERR_MODEL, "onnx: external tensor " + t.name + " without location");
                        auto it = ext_files.find(e.location);
                        if (it == ext_files.end()) {
                            m.files.emplace_back(new MappedFile(base + "/" + e.location));
                            it = ext_files.emplace(e.location, m.files.back().get()).first;
                        }
                        size_t need = (size_t)t.numel() * dtype_size(t.data_type);
                        SDE_REQUIRE(e.offset >= 0 && (size_t)e.offset + need <= it->second->size, SDE_ERR_MODEL,
                                    "onnx: external tensor " + t.name + " exceeds " + e.location);
                        t.data = it->second->data + e.offset; t.nbytes = need;
                    }
                    std::string nm = t.name;
                    m.initializers.emplace(nm, std::move(t));
                    // `owned` storage moved: re-point data for inline-typed tensors
                    OnnxTensor& tt = m.initializers[nm];
                    if (!tt.owned.empty()) tt.data = tt.owned.data();
                } else if (f2 == 11 || f2 == 12) {
                    int wt3; uint64_t v3; Reader s3{}; std::string nm;
                    Reader vi = s2;
                    while (!vi.done()) { int f3 = vi.next(wt3, v3, s3); if (f3 == 1) nm = str_of(s3); }
                    (f2 == 11 ? m.inputs : m.outputs).push_back(nm);
                }
            }
        }
    }
    SDE_REQUIRE(saw_graph, SDE_ERR_MODEL, path + ": no graph in ONNX file");
    for (auto& kv : m.initializers) {
        const OnnxTensor& t = kv.second;
        size_t need = (size_t)t.numel() * dtype_size(t.data_type);
        SDE_REQUIRE(dtype_size(t.data_type) == 0 || t.nbytes >= need, SDE_ERR_MODEL, "onnx: initializer " + kv.first + " is short of data");
    }
}

// ------------------------------------------------------------------------------------------ TOML subset
static std::string trim(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

void pyke_config_load(const std::string& model_dir, PykeConfig& out)
{
    const std::string path = model_dir + "/pyke-diffusers.toml";
    std::ifstream f(path);
    SDE_REQUIRE(f.good(), SDE_ERR_IO, "cannot open " + path);
    std::map<std::string, std::string> kv;
    std::string line, table;
    while (std::getline(f, line)) {
        // strip comments outside strings
        bool in_str = false; size_t cut = std::string::npos;
        for (size_t i = 0; i < line.size(); ++i) { if (line[i] == '"') in_str = !in_str; else if (line[i] == '#' && !in_str) { cut = i; break; } }
        if (cut != std::string::npos) line = line.substr(0, cut);
        line = trim(line);
        if (line.empty()) continue;
        if (line[0] == '[') {
            size_t e = line.find(']');
            SDE_REQUIRE(e != std::string::npos, SDE_ERR_CONFIG, path + ": malformed table header: " + line);
            table = trim(line.substr(1, e - 1));
            continue;
        }
        size_t eq = line.find('=');
        if (eq == std::string::npos) continue;  // continuation lines of multi-line arrays
        std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
        if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
        kv[table.empty() ? k : table + "." + k] = v;
    }
    out.pipeline = kv["pipeline"];
    out.framework_type = kv["framework.type"];
    SDE_REQUIRE(out.pipeline == "stable-diffusion", SDE_ERR_CONFIG, "not a stable diffusion pipeline");   // impl_main.rs:121
    SDE_REQUIRE(out.framework_type == "orte", SDE_ERR_CONFIG, "bad framework");                          // impl_main.rs:113-118
    out.unet_path = kv["unet.path"];
    out.vae_decoder_path = kv["vae.decoder"];
    out.vae_encoder_path = kv["vae.encoder"];
    SDE_REQUIRE(!out.unet_path.empty(), SDE_ERR_CONFIG, path + ": missing [unet].path");
    SDE_REQUIRE(!out.vae_decoder_path.empty(), SDE_ERR_CONFIG, path + ": missing [vae].decoder");
    if (kv.count("vae.scale-factor")) out.vae_scale_factor = atof(kv["vae.scale-factor"].c_str());
    for (auto& e : kv) if (e.first.rfind("hashes.", 0) == 0) out.hashes[e.first.substr(7)] = e.second;
}

// ------------------------------------------------------------------------------------------ weight binding
static std::string module_of(const std::string& node_name)
{
    // "/decoder/mid_block/attentions.0/query/MatMul" -> "decoder.mid_block.attentions.0.query"
    std::vector<std::string> parts;
    std::stringstream ss(node_name); std::string tok;
    while (std::getline(ss, tok, '/')) if (!tok.empty()) parts.push_back(tok);
    std::string out;
    for (size_t i = 0; i + 1 < parts.size(); ++i) { if (i) out += "."; out += parts[i]; }
    return out;
}
static bool ends_with(const std::string& s, const char* suf)
{
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// ------------------------------------------------------------------------------------------ optimizer-fused dialect
// scripts/optimize.py (--preset gpu*) rewrites the exported graphs with onnxruntime's transformer fusions.  Convolutions, plain
// MatMuls, LayerNorm gamma/beta and biases keep their names; what loses its module name is handled here:
//   com.microsoft.Attention      weight [C,3C] = np.stack((q,k,v), axis=1), bias [3C]     (fusion_attention_unet.py:215-226,
//                                                                                          fusion_attention_vae.py:160-190)
//   MatMul_KV_n / MatMul_QKV_n   weight [Cin, heads*2*d] / [C, heads*3*d], per-head interleave   (fusion_attention_unet.py:164-280)
//   com.microsoft.GroupNorm      gamma / beta as <node>_gamma / <node>_beta                       (fusion_group_norm.py:108-160)
//   com.microsoft.FastGelu       tanh-form GELU in place of the exact one                         (fusion_gelu_approximation.py)
//   `*-nv` presets (optimize.py:102-130): NhwcConv keeps the OIHW initializer behind a Transpose (fusion_nhwc_conv.py:62-64, update_weight=False),
//   BiasSplitGelu / BiasAdd / SkipLayerNormalization keep the named bias / gamma / beta initializers: all bound by name, nothing to unpack
// The module a fused node belongs to is read off the first NAMED node that consumes its result (to_out.0 / proj_attn / conv1 ...).
namespace {
struct FusedBinder {
    OnnxModel& m;
    WeightMap& out;
    std::unordered_map<std::string, std::vector<int>> consumers;

    explicit FusedBinder(OnnxModel& m_, WeightMap& o) : m(m_), out(o)
    {
        for (int i = 0; i < (int)m.nodes.size(); ++i)
            for (auto& in : m.nodes[(size_t)i].inputs) consumers[in].push_back(i);
    }
    // module path of a node that still carries the exporter's name; NhwcConv nodes are renamed "NhwcConv_<k>-<old name>"
    // (fusion_nhwc_conv.py:69-75), the path follows the dash
    static std::string path_of(const OnnxNode& n)
    {
        if (n.op_type == "NhwcConv") {
            const size_t dash = n.name.find("-/");
            return dash == std::string::npos ? "" : n.name.substr(dash + 1);
        }
        return n.name;
    }
    static bool named(const OnnxNode& n)
    {
        const std::string p = path_of(n);
        if (p.size() < 2 || p[0] != '/') return false;
        return n.op_type == "MatMul" || n.op_type == "Conv" || n.op_type == "Gemm" || n.op_type == "NhwcConv";
    }
    // module ("a.b.c") of the first named MatMul / Conv / Gemm reachable from `tensor` through unnamed nodes (breadth first)
    std::string consumer_module(const std::string& tensor) const
    {
        std::vector<std::string> frontier{tensor};
        for (int depth = 0; depth < 8 && !frontier.empty(); ++depth) {
            std::vector<std::string> next;
            for (auto& t : frontier) {
                auto it = consumers.find(t);
                if (it == consumers.end()) continue;
                for (int ni : it->second) {
                    const OnnxNode& n = m.nodes[(size_t)ni];
                    if (named(n)) return module_of(path_of(n));
                    for (auto& o : n.outputs) next.push_back(o);
                }
            }
            frontier.swap(next);
        }
        return "";
    }
    static std::string parent(const std::string& mod, const char* suffix)
    {
        size_t n = strlen(suffix);
        return mod.size() > n && mod.compare(mod.size() - n, n, suffix) == 0 ? mod.substr(0, mod.size() - n) : "";
    }
    const OnnxTensor* init(const std::string& name) const
    {
        auto it = m.initializers.find(name);
        return it == m.initializers.end() ? nullptr : &it->second;
    }
    // columns [c0 + g*gstride, +glen) for g = 0..groups-1 of a row-major [rows][cols] tensor -> new [rows][groups*glen] tensor
    const OnnxTensor* take_columns(const OnnxTensor& src, int64_t c0, int64_t glen, int64_t gstride, int64_t groups, const std::string& name)
    {
        SDE_REQUIRE(src.dims.size() == 2 && src.data, SDE_ERR_MODEL, "fused weight " + src.name + " is not a 2-D tensor with data");
        const int64_t rows = src.dims[0], cols = src.dims[1];
        const size_t es = dtype_size(src.data_type);
        SDE_REQUIRE(es && c0 + (groups - 1) * gstride + glen <= cols && src.nbytes >= (size_t)(rows * cols) * es, SDE_ERR_MODEL,
                    "fused weight " + src.name + " has an unexpected shape");
        m.synth.emplace_back();
        OnnxTensor& t = m.synth.back();
        t.name = name; t.data_type = src.data_type; t.dims = {rows, groups * glen};
        t.owned.resize((size_t)(rows * groups * glen) * es);
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t g = 0; g < groups; ++g)
                memcpy(t.owned.data() + (size_t)((r * groups + g) * glen) * es, src.data + (size_t)(r * cols + c0 + g * gstride) * es, (size_t)glen * es);
        t.data = t.owned.data(); t.nbytes = t.owned.size();
        return &t;
    }
    const OnnxTensor* take_range(const OnnxTensor& src, int64_t e0, int64_t n, const std::string& name)
    {
        const size_t es = dtype_size(src.data_type);
        SDE_REQUIRE(es && src.data && src.nbytes >= (size_t)(e0 + n) * es, SDE_ERR_MODEL, "fused bias " + src.name + " is too short");
        m.synth.emplace_back();
        OnnxTensor& t = m.synth.back();
        t.name = name; t.data_type = src.data_type; t.dims = {n};
        t.owned.assign(src.data + (size_t)e0 * es, src.data + (size_t)(e0 + n) * es);
        t.data = t.owned.data(); t.nbytes = t.owned.size();
        return &t;
    }
    static int num_heads(const OnnxNode& n) { const OnnxAttr* a = n.attr("num_heads"); return a ? (int)a->i : 0; }

    void run()
    {
        for (auto& n : m.nodes) {
            const bool ms = n.domain == "com.microsoft";
            if (ms && n.op_type == "FastGelu") { m.fused = m.fast_gelu = true; continue; }
            if (ms && (n.op_type == "Attention" || n.op_type == "MultiHeadAttention" || n.op_type == "GroupNorm")) m.fused = true;
            if (ms && n.op_type == "Attention" && n.inputs.size() >= 2 && !n.outputs.empty()) {
                const OnnxTensor* W = init(n.inputs[1]);
                if (!W) continue;                               // packed-QKV form: the weight sits on MatMul_QKV_n (below)
                const std::string mod = consumer_module(n.outputs[0]);
                std::string scope = parent(mod, ".to_out.0");
                const bool vae = scope.empty();
                if (vae) scope = parent(mod, ".proj_attn");
                SDE_REQUIRE(!scope.empty(), SDE_ERR_MODEL, "fused Attention node " + n.name + ": cannot find the module it belongs to");
                SDE_REQUIRE(W->dims.size() == 2 && W->dims[1] == 3 * W->dims[0], SDE_ERR_MODEL, "fused Attention node " + n.name + ": weight is not [C,3C]");
                const int64_t C = W->dims[0];
                const char* names_u[3] = {".to_q", ".to_k", ".to_v"};
                const char* names_v[3] = {".query", ".key", ".value"};
                const OnnxTensor* B = n.inputs.size() >= 3 ? init(n.inputs[2]) : nullptr;
                for (int j = 0; j < 3; ++j) {
                    const std::string key = scope + (vae ? names_v[j] : names_u[j]);
                    out[key + ".weight"] = BoundTensor{take_columns(*W, j * C, C, 0, 1, key + ".weight"), true};
                    if (vae && B) out[key + ".bias"] = BoundTensor{take_range(*B, j * C, C, key + ".bias"), false};
                }
                if (num_heads(n) > 0) m.fused_heads[scope] = num_heads(n);
            } else if (ms && n.op_type == "MultiHeadAttention" && !n.outputs.empty()) {
                const std::string scope = parent(consumer_module(n.outputs[0]), ".to_out.0");
                if (!scope.empty() && num_heads(n) > 0) m.fused_heads[scope] = num_heads(n);
            } else if (n.op_type == "MatMul" && (n.name.rfind("MatMul_KV", 0) == 0 || n.name.rfind("MatMul_QKV", 0) == 0) && n.inputs.size() == 2) {
                const int parts = n.name.rfind("MatMul_QKV", 0) == 0 ? 3 : 2;
                const OnnxTensor* W = init(n.inputs[1]);
                const OnnxTensor* shp = init(n.name + "_reshape_shape");      // [0, 0, heads, parts, d]
                SDE_REQUIRE(W && shp && shp->data_type == 7 && shp->nbytes >= 40, SDE_ERR_MODEL, "packed " + n.name + ": weight or reshape shape missing");
                int64_t sv[5];
                memcpy(sv, shp->data, 40);
                const int64_t heads = sv[2], d = sv[4];
                SDE_REQUIRE(heads > 0 && d > 0 && sv[3] == parts && W->dims.size() == 2 && W->dims[1] == heads * parts * d, SDE_ERR_MODEL,
                            "packed " + n.name + ": shape does not match [in, heads*" + std::to_string(parts) + "*d]");
                const std::string scope = parent(consumer_module(n.outputs[0]), ".to_out.0");
                SDE_REQUIRE(!scope.empty(), SDE_ERR_MODEL, "packed " + n.name + ": cannot find the attention it feeds");
                const char* names2[2] = {".to_k", ".to_v"};
                const char* names3[3] = {".to_q", ".to_k", ".to_v"};
                for (int j = 0; j < parts; ++j) {
                    const std::string key = scope + (parts == 2 ? names2[j] : names3[j]) + ".weight";
                    out[key] = BoundTensor{take_columns(*W, j * d, d, parts * d, heads, key), true};
                }
                m.fused_heads[scope] = (int)heads;
            } else if (ms && n.op_type == "GroupNorm" && n.inputs.size() >= 3 && !n.outputs.empty()) {
                const OnnxTensor* g = init(n.inputs[1]);
                const OnnxTensor* b = init(n.inputs[2]);
                SDE_REQUIRE(g && b, SDE_ERR_MODEL, "fused GroupNorm node " + n.name + ": gamma / beta are not initializers");
                const std::string mod = consumer_module(n.outputs[0]);
                std::string key;
                std::string s;
                if (!(s = parent(mod, ".conv1")).empty()) key = s + ".norm1";
                else if (!(s = parent(mod, ".conv2")).empty()) key = s + ".norm2";
                else if (!(s = parent(mod, ".proj_in")).empty()) key = s + ".norm";
                else if (!(s = parent(mod, ".proj_attn")).empty()) key = s + ".group_norm";
                else if (!(s = parent(mod, ".query")).empty()) key = s + ".group_norm";
                else if (mod == "conv_out") key = "conv_norm_out";
                else if (!(s = parent(mod, ".conv_out")).empty()) key = s + ".conv_norm_out";
                SDE_REQUIRE(!key.empty(), SDE_ERR_MODEL, "fused GroupNorm node " + n.name + ": unexpected consumer '" + mod + "'");
                out[key + ".weight"] = BoundTensor{g, false};
                out[key + ".bias"] = BoundTensor{b, false};
            }
        }
    }
};
}  // namespace

void bind_weights(OnnxModel& m, WeightMap& out)
{
    std::unordered_map<std::string, std::string> alias;
    for (auto& n : m.nodes)
        if (n.op_type == "Identity" && n.inputs.size() == 1 && n.outputs.size() == 1) alias[n.outputs[0]] = n.inputs[0];
    auto resolve = [&](std::string name) -> const OnnxTensor* {
        for (int hop = 0; hop < 64; ++hop) {
            auto it = alias.find(name);
            if (it == alias.end()) break;
            name = it->second;
        }
        auto it = m.initializers.find(name);
        return it == m.initializers.end() ? nullptr : &it->second;
    };
    for (auto& kv : m.initializers)
        if (ends_with(kv.first, ".weight") || ends_with(kv.first, ".bias")) out[kv.first] = BoundTensor{&kv.second, false};
    for (auto& n : m.nodes) {
        if (n.op_type == "MatMul" && n.inputs.size() == 2 && ends_with(n.name, "/MatMul")) {
            const OnnxTensor* t = resolve(n.inputs[1]);
            if (t && t->dims.size() == 2) out[module_of(n.name) + ".weight"] = BoundTensor{t, true};
        } else if (n.op_type == "Gemm" && n.inputs.size() >= 2) {
            const OnnxTensor* t = resolve(n.inputs[1]);
            const OnnxAttr* tb = n.attr("transB");
            bool transB = tb && tb->i != 0;
            if (t && t->dims.size() == 2 && !transB) out[module_of(n.name) + ".weight"] = BoundTensor{t, true};
        } else if ((n.op_type == "Mul" && ends_with(n.name, "/Mul")) || (n.op_type == "Add" && ends_with(n.name, "/Add"))) {
            for (auto& inp : n.inputs) {
                const OnnxTensor* t = resolve(inp);
                if (t && t->dims.size() == 3 && t->dims[1] == 1 && t->dims[2] == 1)
                    out[module_of(n.name) + (n.op_type == "Mul" ? ".weight" : ".bias")] = BoundTensor{t, false};
            }
        }
    }
    FusedBinder(m, out).run();
}

int onnx_attention_head_dim(const OnnxModel& m, const std::string& scope)
{
    std::unordered_map<std::string, const OnnxTensor*> consts;
    for (auto& n : m.nodes)
        if (n.op_type == "Constant" && !n.outputs.empty()) {
            const OnnxAttr* a = n.attr("value");
            if (a && a->t) consts[n.outputs[0]] = a->t.get();
        }
    for (auto& n : m.nodes) {
        if (n.op_type != "Mul" || n.name.compare(0, scope.size(), scope) != 0) continue;
        for (auto& inp : n.inputs) {
            const OnnxTensor* t = nullptr;
            auto c = consts.find(inp);
            if (c != consts.end()) t = c->second;
            else { auto it = m.initializers.find(inp); if (it != m.initializers.end()) t = &it->second; }
            if (!t || t->numel() != 1 || !t->data) continue;
            float s = 0.f;
            if (t->data_type == 1 && t->nbytes >= 4) memcpy(&s, t->data, 4);
            else if (t->data_type == 10 && t->nbytes >= 2) { half_t h; memcpy(&h, t->data, 2); s = (float)h; }
            else continue;
            if (s > 0.f && s < 1.f) return (int)std::lround(1.0 / ((double)s * s));
        }
    }
    return 0;
}
