"""Seed-compatible random streams: oracle/rng.py against published ChaCha keystreams, and the C ABI
(`sde_rng_*`, host code, no GPU needed) against the oracle, bit for bit.

Parity with the Rust crates' stream (rand 0.8 StdRng + rand_distr 0.4 StandardNormal) is UNPINNED above the ChaCha core:
the reference holds no known-answer vector for it and no Rust toolchain exists here.  Pinned: the ChaCha core."""
import struct

import numpy as np
import pytest

from diffusers_amd import _lib
from oracle import rng as R

# RFC 7539 appendix A.1 test vector #1 (ChaCha20, zero key, zero nonce, counter 0)
CHACHA20_ZERO = ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                 "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
# draft-strombergson-chacha-test-vectors, TC1 (all-zero 256-bit key and IV), 12 rounds, keystream block 1
CHACHA12_ZERO = ("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"
                 "0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")


def _abi_words(key: bytes, rounds: int, n: int) -> np.ndarray:
    out = np.empty(n, np.uint32)
    kb = np.frombuffer(key, np.uint8).copy()
    _lib.check(_lib.lib().sde_rng_chacha_words(kb.ctypes.data, rounds, out.ctypes.data, n))
    return out


def _abi_normal(seed: int, skip: int, n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    _lib.check(_lib.lib().sde_rng_standard_normal(seed, skip, out.ctypes.data, n))
    return out


@pytest.mark.parametrize("rounds,kat", [(20, CHACHA20_ZERO), (12, CHACHA12_ZERO)])
def test_chacha_core_known_answers(rounds, kat):
    assert struct.pack("<16I", *R.chacha_block([0] * 8, 0, rounds)).hex() == kat
    assert _abi_words(bytes(32), rounds, 16).tobytes().hex() == kat


def test_chacha_stream_abi_matches_oracle_across_refills():
    key = bytes(range(32))
    rng = R.StdRng(key, 12)
    ref = np.array([rng.next_u32() for _ in range(64 * 3 + 5)], np.uint32)   # three 4-block refills, counter 0..11
    assert np.array_equal(_abi_words(key, 12, ref.size), ref)
    # the block counter is the only thing that changes between blocks
    assert R.chacha_block(struct.unpack("<8I", key), 5, 12) == list(ref[80:96])


def test_next_u64_is_two_words_low_first_and_straddles_refills():
    a, b = R.StdRng(bytes(32), 12), R.StdRng(bytes(32), 12)
    words = [a.next_u32() for _ in range(130)]
    assert b.next_u64() == words[0] | (words[1] << 32)
    b.next_u32()                                   # odd index: the value at words 63/64 straddles the refill
    for _ in range(30):
        b.next_u64()
    assert b.index == 63
    assert b.next_u64() == words[63] | (words[64] << 32)
    assert b.next_u64() == words[65] | (words[66] << 32)


@pytest.mark.parametrize("seed", [0, 1, 42, 31337, 2**64 - 1])
def test_seed_expansion_abi_matches_oracle(seed):
    out = np.empty(32, np.uint8)
    _lib.check(_lib.lib().sde_rng_seed_key(seed, out.ctypes.data))
    assert out.tobytes() == R.seed_key(seed)
    assert R.seed_key(seed) != R.seed_key((seed + 1) % 2**64)


def test_ziggurat_tables():
    x, f = R.ZIG_NORM_X, R.ZIG_NORM_F
    assert len(x) == 257 and x[1] == R.ZIG_NORM_R and x[256] == 0.0 and f[256] == 1.0
    assert all(x[i] > x[i + 1] for i in range(256))
    # every layer has the same area V: x[i] * (f[i+1] - f[i]) == V for the rectangles
    for i in range(1, 255):
        assert abs(x[i] * (f[i + 1] - f[i]) - 0.00492867323399) < 1e-14
    assert abs(x[0] - 3.91075795953709) < 1e-13


@pytest.mark.parametrize("seed,n", [(0, 4096), (42, 4096), (2**63 + 12345, 4096), (7, 60000)])
def test_standard_normal_abi_matches_oracle_bit_for_bit(seed, n):
    """n = 60000 crosses ~15 tail draws (|x| > R) and ~700 wedge tests, the two paths that consume extra words."""
    ref = R.random_using((n,), R.StdRng.seed_from_u64(seed))
    got = _abi_normal(seed, 0, n)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    if n >= 60000:
        assert (np.abs(ref) > R.ZIG_NORM_R).sum() >= 5
        assert abs(ref.mean()) < 0.02 and abs(ref.std() - 1) < 0.02
        assert abs((np.abs(ref) < 1).mean() - 0.6827) < 0.01


def test_skip_continues_the_stream():
    full = _abi_normal(9, 0, 3000)
    assert np.array_equal(_abi_normal(9, 1024, 1976), full[1024:])
    assert _abi_normal(9, 0, 0).size == 0


def test_latent_and_scheduler_streams_differ():
    # impl_txt2img.rs:249 vs :267: seed and seed + 31337
    assert not np.array_equal(_abi_normal(5, 0, 64), _abi_normal(5 + 31337, 0, 64))
