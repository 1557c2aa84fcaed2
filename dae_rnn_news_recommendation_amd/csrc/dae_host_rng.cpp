// dae_host_rng.cpp -- host side of the reference-exact corruption mode (rng = "numpy").
//
// The reference draws its masking noise from NumPy's legacy global RandomState: `np.random.rand(nnz) >= v` over the stored
// entries of the WHOLE train set once per epoch (autoencoder/utils.py:111), i.e. 2 MT19937 words per stored entry -- 3.2 M words
// per epoch at 8000 x 10000 x 2 %.  NumPy produces them one double at a time (~2.5 ns each, 4 ms per epoch) and that host draw,
// not the GPU (2 ms per epoch), bounded fit(rng="numpy").  This file continues the SAME stream natively: it takes the
// RandomState's 624-word state and position, regenerates the state block-wise with the three dependency-free phases of the
// MT19937 recurrence (vectorisable: AVX2 / AVX-512 by function multiversioning), tempers, and turns word pairs straight into
// packed keep bits with an exact integer comparison -- no doubles, no bool array, no packbits pass:
//     rand() = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 >= v    <=>    ((a >> 5) << 26 | (b >> 6)) >= ceil(v * 2^53)
// (both sides are exact: the numerator is a 53-bit integer and v * 2^53 is a power-of-two scaling of the double v).
// The caller writes the advanced state back with np.random.set_state, so every later legacy draw (the epoch's shuffle, the
// next epoch's noise) continues as if NumPy itself had drawn the numbers.  Pinned bit-for-bit against NumPy in
// tests/test_host_rng.py.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/dae_hip.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;

#define DAE_MT_PHASE(lo, hi, off)                                                   \
    _Pragma("clang loop vectorize(assume_safety)")                                 \
    for (int k = (lo); k < (hi); ++k) {                                             \
        const uint32_t y = (mt[k] & UPPER) | (mt[k + 1] & LOWER);                   \
        mt[k] = mt[k + (off)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);              \
    }

// next 624 raw words (genrand's block step), then the tempered copy in out[]
#define DAE_MT_BODY                                                                 \
    DAE_MT_PHASE(0, MT_N - MT_M, MT_M)              /* reads old [397, 624) */      \
    DAE_MT_PHASE(MT_N - MT_M, 2 * (MT_N - MT_M), MT_M - MT_N) /* reads new [0, 227) */ \
    DAE_MT_PHASE(2 * (MT_N - MT_M), MT_N - 1, MT_M - MT_N)    /* reads new [227, 396) */ \
    {                                                                               \
        const uint32_t y = (mt[MT_N - 1] & UPPER) | (mt[0] & LOWER);                \
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);        \
    }                                                                               \
    _Pragma("clang loop vectorize(enable)")                                        \
    for (int k = 0; k < MT_N; ++k) {                                                \
        uint32_t y = mt[k];                                                         \
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18; \
        out[k] = y;                                                                 \
    }

__attribute__((target("avx512f,avx512bw,avx512vl"))) void regen_avx512(uint32_t* __restrict__ mt, uint32_t* __restrict__ out) { DAE_MT_BODY }
__attribute__((target("avx2"))) void regen_avx2(uint32_t* __restrict__ mt, uint32_t* __restrict__ out) { DAE_MT_BODY }
void regen_base(uint32_t* __restrict__ mt, uint32_t* __restrict__ out) { DAE_MT_BODY }

// keep bits of `np` (<= 32) consecutive (a, b) word pairs starting at w[0]; bit j of the result = pair j
#define DAE_KEEP_BODY                                                               \
    uint32_t word = 0;                                                              \
    _Pragma("clang loop vectorize(enable)")                                        \
    for (int j = 0; j < np; ++j) {                                                  \
        const uint64_t k = ((uint64_t)(w[2 * j] >> 5) << 26) | (uint64_t)(w[2 * j + 1] >> 6); \
        word |= (uint32_t)(k >= thr) << j;                                          \
    }                                                                               \
    return word;

__attribute__((target("avx512f,avx512bw,avx512vl,avx512dq"))) uint32_t keep_avx512(const uint32_t* __restrict__ w, int np, uint64_t thr) { DAE_KEEP_BODY }
__attribute__((target("avx2"))) uint32_t keep_avx2(const uint32_t* __restrict__ w, int np, uint64_t thr) { DAE_KEEP_BODY }
uint32_t keep_base(const uint32_t* __restrict__ w, int np, uint64_t thr) { DAE_KEEP_BODY }

typedef void (*regen_fn)(uint32_t*, uint32_t*);
typedef uint32_t (*keep_fn)(const uint32_t*, int, uint64_t);

void pick(regen_fn& rg, keep_fn& kp) {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
        __builtin_cpu_supports("avx512dq")) { rg = regen_avx512; kp = keep_avx512; }
    else if (__builtin_cpu_supports("avx2")) { rg = regen_avx2; kp = keep_avx2; }
    else { rg = regen_base; kp = keep_base; }
}

}  // namespace

// See include/dae_hip.h.  key[624] / pos are NumPy's ('MT19937', key, pos, ...) state fields; both are advanced in place.
extern "C" int dae_host_mt19937_keep_bits(uint32_t* key, int32_t* pos_io, int64_t n, double corr_frac, uint32_t* bits_out) {
    if (!key || !pos_io || !bits_out || n < 0 || *pos_io < 0 || *pos_io > MT_N || !(corr_frac >= 0.0 && corr_frac <= 1.0)) return 1;
    static regen_fn rg = nullptr;
    static keep_fn kp = nullptr;
    if (!rg) pick(rg, kp);
    // rand() >= v  <=>  53-bit numerator >= ceil(v * 2^53); v = 0 keeps everything, v = 1 nothing (numerator < 2^53)
    const uint64_t thr = (uint64_t)ceil(ldexp(corr_frac, 53));
    alignas(64) uint32_t out[MT_N];                   // tempered words of the current block
    int pos = *pos_io;
    // NumPy keeps the RAW state; the tempered words of the unconsumed part [pos, 624) of the current block are re-derived here
    for (int k = 0; k < MT_N; ++k) {
        uint32_t y = key[k];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        out[k] = y;
    }
    const int64_t n_words = (n + 31) / 32;
    memset(bits_out, 0, (size_t)n_words * 4);
    int64_t i = 0;                                    // entries done
    while (i < n) {
        const int64_t left = n - i;
        int np = (MT_N - pos) / 2;                    // whole pairs left in this block
        if (np > 32) np = 32;
        if (np > left) np = (int)left;
        if (np > 0) {
            const uint32_t m = kp(out + pos, np, thr);
            const int sh = (int)(i & 31);
            bits_out[i >> 5] |= m << sh;
            if (sh && (m >> (32 - sh))) bits_out[(i >> 5) + 1] |= m >> (32 - sh);   // only set bits are carried: stays inside n_words
            pos += 2 * np; i += np;
            continue;
        }
        // block boundary: 0 or 1 word left -- draw this entry's two words one by one
        uint32_t ab[2];
        for (int h = 0; h < 2; ++h) {
            if (pos >= MT_N) { rg(key, out); pos = 0; }
            ab[h] = out[pos++];
        }
        const uint64_t k = ((uint64_t)(ab[0] >> 5) << 26) | (uint64_t)(ab[1] >> 6);
        bits_out[i >> 5] |= (uint32_t)(k >= thr) << (i & 31);
        ++i;
    }
    *pos_io = pos;
    return 0;
}
