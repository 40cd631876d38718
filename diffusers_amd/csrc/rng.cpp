// Ziggurat StandardNormal over the ChaCha12 stream + the C ABI of the seed-compatible RNG (see rng.h for provenance).
// Built with -ffp-contract=off: the Rust crates never fuse a multiply-add, and one fused rounding can flip an accept/reject test.
#include "rng.h"

#include <cmath>
#include <string>

#include "../../include/sdengine.h"
#include "common.h"

namespace sde {

struct ZigguratNormal {
    static constexpr double R = 3.654152885361009;   // ZIG_NORM_R
    double x[257], f[257];
    ZigguratNormal()
    {
        const double v = 0.00492867323399;            // area of each layer (unnormalised pdf exp(-x^2/2))
        x[0] = v / std::exp(-0.5 * R * R);
        x[1] = R;
        for (int i = 2; i < 256; ++i) x[i] = std::sqrt(-2.0 * std::log(v / x[i - 1] + std::exp(-0.5 * x[i - 1] * x[i - 1])));
        x[256] = 0.0;
        for (int i = 0; i < 257; ++i) f[i] = std::exp(-0.5 * x[i] * x[i]);
    }
    static inline double bits_to_f64(uint64_t frac52, int exponent)
    {
        const uint64_t b = ((uint64_t)(exponent + 1023) << 52) | frac52;
        double d;
        memcpy(&d, &b, 8);
        return d;
    }
    static inline double open01(ChaChaRng& rng) { return bits_to_f64(rng.next_u64() >> 12, 0) - (1.0 - 2.220446049250313e-16 / 2.0); }
    static inline double standard_f64(ChaChaRng& rng) { return (double)(rng.next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    double sample(ChaChaRng& rng) const
    {
        for (;;) {
            const uint64_t bits = rng.next_u64();
            const int i = (int)(bits & 0xff);
            const double u = bits_to_f64(bits >> 12, 1) - 3.0;    // [-1, 1)
            const double v = u * x[i];
            if (std::fabs(v) < x[i + 1]) return v;
            if (i == 0) {                                             // tail beyond R
                double a = 1.0, b = 0.0;
                while (-2.0 * b < a * a) {
                    const double a_ = open01(rng), b_ = open01(rng);
                    a = std::log(a_) / R;
                    b = std::log(b_);
                }
                return u < 0.0 ? a - R : R - a;
            }
            if (f[i + 1] + (f[i] - f[i + 1]) * standard_f64(rng) < std::exp(-0.5 * v * v)) return v;
        }
    }
};

static const ZigguratNormal& ziggurat_normal()
{
    static const ZigguratNormal z;
    return z;
}

void fill_standard_normal(ChaChaRng& rng, float* dst, size_t n)
{
    const ZigguratNormal& z = ziggurat_normal();
    for (size_t i = 0; i < n; ++i) dst[i] = (float)z.sample(rng);
}

}  // namespace sde

using namespace sde;

sde_status sde_rng_standard_normal(uint64_t seed, uint64_t skip, float* out, size_t n)
{
    return guarded([&] {
        SDE_REQUIRE(out || n == 0, SDE_ERR_INVALID_ARGUMENT, "null output");
        ChaChaRng rng = ChaChaRng::seed_from_u64(seed);
        const ZigguratNormal& z = ziggurat_normal();
        for (uint64_t i = 0; i < skip; ++i) (void)z.sample(rng);
        fill_standard_normal(rng, out, n);
    });
}

sde_status sde_rng_chacha_words(const uint8_t key[32], int rounds, uint32_t* out, size_t n_words)
{
    return guarded([&] {
        SDE_REQUIRE(key && (out || n_words == 0), SDE_ERR_INVALID_ARGUMENT, "null argument");
        SDE_REQUIRE(rounds > 0 && rounds % 2 == 0, SDE_ERR_INVALID_ARGUMENT, "rounds must be even and positive");
        ChaChaRng rng = ChaChaRng::from_key(key, rounds);
        for (size_t i = 0; i < n_words; ++i) out[i] = rng.next_u32();
    });
}

sde_status sde_rng_seed_key(uint64_t seed, uint8_t key_out[32])
{
    return guarded([&] {
        SDE_REQUIRE(key_out, SDE_ERR_INVALID_ARGUMENT, "null argument");
        ChaChaRng rng = ChaChaRng::seed_from_u64(seed);
        for (int i = 0; i < 8; ++i) for (int b = 0; b < 4; ++b) key_out[4 * i + b] = (uint8_t)(rng.key[i] >> (8 * b));
    });
}
