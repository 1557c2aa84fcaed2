// dae_sym.h -- Gs = scale * (G + G^T) on 64 x 64 tiles (autodiff of D = h h^T), shared by the stand-alone kernel
// (dae_elementwise.hip) and the extra workgroups of the decode launch (dae_gemm.hip).
#pragma once
#include "dae_common.h"

namespace dae {

// one 64 x 64 tile (bx, by) of Gs; 256 threads; tile = 64 x 65 floats of LDS
template <typename T>
__device__ __forceinline__ void sym_scale_tile(const float* __restrict__ G, int B, int Bp, const float* __restrict__ tri_scalars,
                                               T* __restrict__ Gs, int bx, int by, float (*tile)[65], float mul = 1.f) {
    const int j0 = bx * 64, i0 = by * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // all 32 loads of a thread are issued before the first use (pure latency: 196 tiles, 3 MB)
    float gt[16], gd[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = ty + 4 * k;
        const int a = j0 + r, b = i0 + tx;             // tile of G^T: element (j0+r, i0+tx)
        gt[k] = (a < B && b < B) ? G[(int64_t)a * Bp + b] : 0.f;
        const int i = i0 + r, j = j0 + tx;
        gd[k] = (i < B && j < B) ? G[(int64_t)i * Bp + j] : 0.f;
    }
    const float sc = tri_scalars[0] * mul;        // mul: operand scale of the 16-bit modes (a power of two; 1 otherwise)
#pragma unroll
    for (int k = 0; k < 16; ++k) tile[ty + 4 * k][tx] = gt[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = ty + 4 * k;
        const int i = i0 + r, j = j0 + tx;
        float v = 0.f;
        if (i < B && j < B) v = sc * (gd[k] + tile[tx][r]);
        if constexpr (sizeof(T) == 2) v = sat16(v);
        Gs[(int64_t)i * Bp + j] = Elem<T>::from(v);
    }
}

}  // namespace dae
