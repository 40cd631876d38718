// loader.h — model directory loading: pyke-diffusers.toml, ONNX protobuf reader, weight binding.
// Replaces what `SessionBuilder::with_model_from_file` does for the reference
// (/root/reference/src/pipelines/stable_diffusion/impl_main.rs:136-157) for the two graphs on the hot path.
#pragma once
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

// ---- memory-mapped file
struct MappedFile {
    const uint8_t* data = nullptr;
    size_t size = 0;
    int fd = -1;
    explicit MappedFile(const std::string& path);
    ~MappedFile();
    MappedFile(const MappedFile&) = delete;
};

// ---- ONNX (subset; field numbers in SURVEY.md Appendix C)
struct OnnxTensor {
    std::string name;
    std::vector<int64_t> dims;
    int data_type = 0;            // 1=f32 7=i64 10=f16 ...
    const uint8_t* data = nullptr;  // points into a mapped file or into `owned`
    size_t nbytes = 0;
    std::vector<uint8_t> owned;
    int64_t numel() const { int64_t n = 1; for (auto d : dims) n *= d; return n; }
};
struct OnnxAttr {
    std::string name;
    bool has_f = false, has_i = false;
    float f = 0; int64_t i = 0;
    std::string s;
    std::vector<int64_t> ints;
    std::shared_ptr<OnnxTensor> t;
};
struct OnnxNode {
    std::string op_type, name, domain;
    std::vector<std::string> inputs, outputs;
    std::vector<OnnxAttr> attrs;
    const OnnxAttr* attr(const char* n) const { for (auto& a : attrs) if (a.name == n) return &a; return nullptr; }
};
struct OnnxModel {
    int64_t ir_version = 0;
    std::string producer;
    std::map<std::string, int64_t> opsets;
    std::vector<OnnxNode> nodes;
    std::unordered_map<std::string, OnnxTensor> initializers;
    std::vector<std::string> inputs, outputs;
    std::vector<std::unique_ptr<MappedFile>> files;  // keep mappings alive
    // filled by bind_weights for the optimizer-fused dialect (scripts/optimize.py): per-module tensors unpacked from the packed
    // Attention / MatMul_KV weights (std::deque: stable addresses), head counts of the fused attention nodes by module scope,
    // and whether the feed-forward GELU was replaced by com.microsoft.FastGelu (tanh form)
    std::deque<OnnxTensor> synth;
    std::map<std::string, int> fused_heads;
    bool fused = false, fast_gelu = false;
};
void onnx_load(const std::string& path, OnnxModel& out);

// ---- pyke-diffusers.toml (config.rs:17-117) — only the keys the hot path needs
struct PykeConfig {
    std::string pipeline, framework_type;
    std::string unet_path, vae_decoder_path, vae_encoder_path;
    double vae_scale_factor = 0.18215;  // parsed, and — like the reference (impl_main.rs:399) — not used
    std::map<std::string, std::string> hashes;
};
void pyke_config_load(const std::string& model_dir, PykeConfig& out);

// ---- bound weights: diffusers parameter name -> tensor view
struct BoundTensor {
    const OnnxTensor* t = nullptr;
    bool in_out = false;       // stored [in][out] (onnx::MatMul_n) rather than [out][in]
    int64_t out_dim() const { return in_out ? t->dims[1] : t->dims[0]; }
    int64_t in_dim() const { return in_out ? t->dims[0] : t->dims[1]; }
};
typedef std::unordered_map<std::string, BoundTensor> WeightMap;
void bind_weights(OnnxModel& m, WeightMap& out);
// head dim of the first attention whose node names start with `scope` (from the softmax scale constant); 0 if absent
int onnx_attention_head_dim(const OnnxModel& m, const std::string& scope);
