"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product).

CPU restatement of the random streams the reference's txt2img loop draws from:

    let mut rng = StdRng::seed_from_u64(seed);                                   impl_txt2img.rs:249
    Array4::<f32>::random_using(latents_shape, StandardNormal, &mut rng)         impl_txt2img.rs:262
    let mut scheduler_rng = StdRng::seed_from_u64(seed + 31337);                 impl_txt2img.rs:267
    Array4::<f32>::random_using(model_output.raw_dim(), StandardNormal, rng)     euler_ancestral_discrete.rs:203, ddpm.rs:224,
                                                                                 euler_discrete.rs:200 (drawn, unused at gamma = 0)

The arithmetic lives in crates the reference does not vendor (`ndarray-rand = "0.14"`, Cargo.toml:23 -> rand 0.8,
rand_core 0.6, rand_chacha 0.3, rand_distr 0.4); it is restated from their published algorithms:
seed_from_u64 = PCG32 key expansion; StdRng = ChaCha12 (64-bit counter, stream 0, 4-block buffer, next_u64 = two
consecutive words, low first); StandardNormal = 256-layer Ziggurat on f64 (R = 3.654152885361009, V = 0.00492867323399,
x[0] = V/f(R), x[i+1] = f^-1(V/x[i] + f(x[i])), x[256] = 0), sampled as f64 and cast to f32.

PARITY UNPINNED above the ChaCha core: no Rust toolchain and no known-answer stream of those crates is available here.
The ChaCha core is pinned by the published zero-key keystreams (ChaCha20: RFC 7539 A.1 #1; ChaCha12: the eSTREAM /
draft-strombergson-chacha-test-vectors TC1, 256-bit key), see tests/test_rng.py.
"""
from __future__ import annotations

import math
import struct

import numpy as np

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def _rotl(v: int, n: int) -> int:
    return ((v << n) & M32) | (v >> (32 - n))


def chacha_block(key_words, counter: int, rounds: int):
    """One 16-word ChaCha block: constants | key | 64-bit counter | 64-bit stream id (0)."""
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574, *key_words, counter & M32, (counter >> 32) & M32, 0, 0]
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, init)]


def seed_key(seed: int) -> bytes:
    """rand_core 0.6 `SeedableRng::seed_from_u64`: PCG32 output, 4 key bytes per draw."""
    state = seed & M64
    out = b""
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32
        out += struct.pack("<I", x)
    return out


class StdRng:
    """rand 0.8 `StdRng` = ChaCha12 behind rand_core's BlockRng (64-word buffer)."""

    def __init__(self, key: bytes, rounds: int = 12):
        self.key = struct.unpack("<8I", key)
        self.rounds = rounds
        self.counter = 0
        self.buf = []
        self.index = 64

    @classmethod
    def seed_from_u64(cls, seed: int) -> "StdRng":
        return cls(seed_key(seed), 12)

    def _refill(self):
        self.buf = []
        for b in range(4):
            self.buf += chacha_block(self.key, self.counter + b, self.rounds)
        self.counter += 4

    def next_u32(self) -> int:
        if self.index >= 64:
            self._refill(); self.index = 0
        v = self.buf[self.index]
        self.index += 1
        return v

    def next_u64(self) -> int:
        i = self.index
        if i < 63:
            self.index += 2
            return (self.buf[i + 1] << 32) | self.buf[i]
        if i >= 64:
            self._refill(); self.index = 2
            return (self.buf[1] << 32) | self.buf[0]
        lo = self.buf[63]
        self._refill(); self.index = 1
        return (self.buf[0] << 32) | lo


ZIG_NORM_R = 3.654152885361009
_V = 0.00492867323399


def _tables():
    f = lambda t: math.exp(-t * t / 2.0)
    x = [0.0] * 257
    x[0] = _V / f(ZIG_NORM_R)
    x[1] = ZIG_NORM_R
    for i in range(2, 256):
        x[i] = math.sqrt(-2.0 * math.log(_V / x[i - 1] + f(x[i - 1])))
    x[256] = 0.0
    return x, [f(t) for t in x]


ZIG_NORM_X, ZIG_NORM_F = _tables()


def _float_with_exponent(frac52: int, exponent: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", ((exponent + 1023) << 52) | frac52))[0]


def _open01(rng: StdRng) -> float:
    return _float_with_exponent(rng.next_u64() >> 12, 0) - (1.0 - 2.220446049250313e-16 / 2.0)


def standard_normal_f64(rng: StdRng) -> float:
    """rand_distr 0.4 `StandardNormal` (utils::ziggurat, symmetric)."""
    while True:
        bits = rng.next_u64()
        i = bits & 0xFF
        u = _float_with_exponent(bits >> 12, 1) - 3.0
        x = u * ZIG_NORM_X[i]
        if abs(x) < ZIG_NORM_X[i + 1]:
            return x
        if i == 0:
            a, b = 1.0, 0.0
            while -2.0 * b < a * a:
                a_ = _open01(rng); b_ = _open01(rng)
                a = math.log(a_) / ZIG_NORM_R
                b = math.log(b_)
            return a - ZIG_NORM_R if u < 0.0 else ZIG_NORM_R - a
        if ZIG_NORM_F[i + 1] + (ZIG_NORM_F[i] - ZIG_NORM_F[i + 1]) * ((rng.next_u64() >> 11) * (1.0 / 9007199254740992.0)) < math.exp(-x * x / 2.0):
            return x


def random_using(shape, rng: StdRng) -> np.ndarray:
    """`ArrayN::<f32>::random_using(shape, StandardNormal, rng)`: row-major fill, one f64 sample cast to f32 per element."""
    n = int(np.prod(shape))
    return np.array([standard_normal_f64(rng) for _ in range(n)], dtype=np.float64).astype(np.float32).reshape(shape)
