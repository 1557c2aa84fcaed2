// dae_rng.h -- Philox4x32-10 counter RNG (Salmon et al., SC'11) for on-device input corruption.
// The reference draws its masking noise from NumPy's legacy MT19937 stream on the host
// (utils.py:108,111); that stream is reproduced bit-exactly by the host path (keep-bit upload).
// This device RNG is the statistically-equivalent fast path: element e of the train set is kept
// iff philox_uniform(e; seed, stream) >= corr_frac, independent of batch order.  oracle/dae_oracle.py
// (philox_uniform) restates it for the tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dae {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0; k.y += W1;
    }
    return c;
}

// uniform in [0,1) with 24 bits; counter = (idx_lo, idx_hi, stream, 0), key = (seed_lo, seed_hi)
__device__ __forceinline__ float philox_uniform(uint64_t idx, uint64_t seed, uint32_t stream) {
    uint4 c = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u);
    uint2 k = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    uint4 o = philox4x32_10(c, k);
    return (float)(o.x >> 8) * 5.9604644775390625e-8f;   // 2^-24
}

// Dense-ndarray masking: ONE Philox evaluation serves four neighbouring elements.  Element (row, f) of the train set takes
// word f & 3 of the draw at counter (f >> 2, row, stream, 2), key = seed  (oracle.philox_uniform_dense restates it).
__device__ __forceinline__ uint4 philox_dense4(uint32_t row, uint32_t f_quad, uint64_t seed, uint32_t stream) {
    return philox4x32_10(make_uint4(f_quad, row, stream, 2u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}
__device__ __forceinline__ float philox_word_uniform(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-8f; }

}  // namespace dae
