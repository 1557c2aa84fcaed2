// dae_gather.hip -- corrupt + gather the mini-batch rows of the resident train set into dense
// MFMA operand tiles (front half of K0/K1).
//
// Replaces, per batch: utils.masking_noise (utils.py:94-115; applied once per epoch to the whole set
// in the reference, here evaluated lazily per gathered entry with the SAME per-entry keep decision),
// the CSR fancy-index (utils.py:59-60), get_sparse_ind_val_shape (utils.py:162-180) and
// tf.sparse.to_dense (triplet_loss_utils.py:264).
//
// CSR path: one workgroup per (batch row, 4096-column chunk).  The row's entries falling in the chunk
// are a contiguous run of the (sorted) CSR row, found by two binary searches; the run is read with
// coalesced 4-byte loads, scattered into two zero-initialised LDS row tiles (clean x and corrupted x~)
// and the tiles are streamed out with 16-byte coalesced stores.  Only kept entries touch the
// (pre-zeroed) transposed operand x~^T.
#include "dae_common.h"
#include "dae_label.h"
#include "dae_rng.h"

namespace dae {

constexpr int GATHER_THREADS = 256;
constexpr int GATHER_CW = 4096;   // columns per workgroup

__device__ __forceinline__ bool keep_entry(int mode, const uint32_t* bits, uint64_t e, uint64_t seed, uint32_t stream,
                                           float frac) {
    if (mode == DAE_CORR_KEEPBITS) return (bits[e >> 5] >> (e & 31)) & 1u;
    if (mode == DAE_CORR_PHILOX_MASK) return philox_uniform(e, seed, stream) >= frac;
    return true;
}

template <typename T>
__global__ __launch_bounds__(GATHER_THREADS) void gather_csr_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int32_t* __restrict__ row_idx, int B, int F, T* __restrict__ x, T* __restrict__ xc, int64_t ldx,
    T* __restrict__ xct, int64_t ldt, float* __restrict__ rowsq, int corr_mode, const uint32_t* __restrict__ keep_bits,
    uint64_t seed, uint32_t stream, float corr_frac, float scale, uint32_t* __restrict__ xc_bits, int64_t ldw,
    LabelJob job, int label_slice, uint32_t* __restrict__ x_bits) {
    // row tiles; the same LDS serves the label-statistics block (blockIdx.y == label_slice, blockIdx.x == 0)
    constexpr int TILE_B = 2 * GATHER_CW * (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) char smem_raw[TILE_B > LABEL_SMEM_BYTES ? TILE_B : LABEL_SMEM_BYTES];
    __shared__ uint32_t lbits[GATHER_CW / 32];
    __shared__ uint32_t lbits_x[GATHER_CW / 32];
    __shared__ float red[GATHER_THREADS / 64];
    if ((int)blockIdx.y == label_slice) {
        if (blockIdx.x == 0) label_stats_block<GATHER_THREADS>(job, smem_raw);
        return;
    }
    T* lx = reinterpret_cast<T*>(smem_raw);
    T* lxc = lx + GATHER_CW;
    const int i = blockIdx.x;            // batch row (0..Bp-1)
    const int c0 = blockIdx.y * GATHER_CW;
    const int tid = threadIdx.x;
    const int ncol = min(GATHER_CW, (int)ldx - c0);   // ldx = Fp
    constexpr int VEC = 16 / sizeof(T);
    // zero the tiles
    for (int k = tid * VEC; k < GATHER_CW; k += GATHER_THREADS * VEC) {
        if (x) *reinterpret_cast<i32x4*>(&lx[k]) = i32x4{0, 0, 0, 0};
        if (xc) *reinterpret_cast<i32x4*>(&lxc[k]) = i32x4{0, 0, 0, 0};
    }
    if (tid < GATHER_CW / 32) { lbits[tid] = 0u; lbits_x[tid] = 0u; }
    __syncthreads();
    if (i < B) {
        const int64_t row = row_idx[i];
        const int64_t s = indptr[row], e = indptr[row + 1];
        // first entry with column >= c0 / >= c0 + GATHER_CW (uniform binary searches)
        int64_t lo = s, hi = e;
        {
            int64_t a = s, b = e;
            while (a < b) { int64_t m = (a + b) >> 1; if (indices[m] < c0) a = m + 1; else b = m; }
            lo = a;
            b = e;
            const int cend = c0 + GATHER_CW;
            while (a < b) { int64_t m = (a + b) >> 1; if (indices[m] < cend) a = m + 1; else b = m; }
            hi = a;
        }
        for (int64_t k = lo + tid; k < hi; k += GATHER_THREADS) {
            const int col = indices[k];
            const float v = values ? values[k] : 1.0f;
            const bool keep = keep_entry(corr_mode, keep_bits, (uint64_t)k, seed, stream, corr_frac);
            const float vc = keep ? v * scale : 0.0f;
            if (col < F) {
                if (x) lx[col - c0] = Elem<T>::from(v);
                if (xc) lxc[col - c0] = Elem<T>::from(vc);
                if (xc_bits && keep) atomicOr(&lbits[(col - c0) >> 5], 1u << ((col - c0) & 31));
                if (x_bits) atomicOr(&lbits_x[(col - c0) >> 5], 1u << ((col - c0) & 31));
                if (xct && keep) xct[(int64_t)col * ldt + i] = Elem<T>::from(vc);
            }
        }
        if (rowsq && blockIdx.y == 0) {   // sum of squares of the whole clean row (cosine_proximity)
            float acc = 0.f;
            for (int64_t k = s + tid; k < e; k += GATHER_THREADS) {
                const float v = values ? values[k] : 1.0f;
                acc += v * v;
            }
            acc = wave_sum(acc);
            if ((tid & 63) == 0) red[tid >> 6] = acc;
        }
    }
    __syncthreads();
    if (rowsq && blockIdx.y == 0 && tid == 0) {
        float t = 0.f;
        if (i < B) for (int w = 0; w < GATHER_THREADS / 64; ++w) t += red[w];
        rowsq[i] = t;
    }
    for (int k = tid * VEC; k < ncol; k += GATHER_THREADS * VEC) {
        if (x) *reinterpret_cast<i32x4*>(&x[(int64_t)i * ldx + c0 + k]) = *reinterpret_cast<const i32x4*>(&lx[k]);
        if (xc) *reinterpret_cast<i32x4*>(&xc[(int64_t)i * ldx + c0 + k]) = *reinterpret_cast<const i32x4*>(&lxc[k]);
    }
    // bit-packed x~ (binary inputs): bit b of word w of row i <=> feature 32*w + b is kept.  Operand of gemm_encode_bits.
    if (xc_bits && tid < (ncol >> 5)) xc_bits[(int64_t)i * ldw + (c0 >> 5) + tid] = lbits[tid];
    // bit image of the CLEAN row (binary inputs): what the decode epilogue reads instead of the dense x tile
    if (x_bits && tid < (ncol >> 5)) x_bits[(int64_t)i * ldw + (c0 >> 5) + tid] = lbits_x[tid];
}

// Dense ndarray input (autoencoder.py:143 sparse_input False; dense masking utils.py:107-109).
// 64x64 tiles: coalesced fp32 reads along f, LDS transpose for the x~^T operand.
template <typename T>
__global__ __launch_bounds__(256) void gather_dense_kernel(
    const float* __restrict__ data, int64_t ld_data, const int32_t* __restrict__ row_idx, int B, int F,
    T* __restrict__ x, T* __restrict__ xc, int64_t ldx, T* __restrict__ xct, int64_t ldt, float* __restrict__ rowsq_part,
    int corr_mode, const uint32_t* __restrict__ keep_bits, uint64_t seed, uint32_t stream, float corr_frac, float scale) {
    __shared__ float tile[64][65];
    __shared__ float sq[4][64];
    const int i0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // ty in 0..3
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int i = i0 + r, f = f0 + tx;
        float v = 0.f, vc = 0.f;
        if (i < B && f < F) {
            const int64_t row = row_idx[i];
            v = data[row * ld_data + f];
            const uint64_t eidx = (uint64_t)row * (uint64_t)F + (uint64_t)f;
            const bool keep = keep_entry(corr_mode, keep_bits, eidx, seed, stream, corr_frac);
            vc = keep ? v * scale : 0.f;
        }
        if (x) x[(int64_t)i * ldx + f] = Elem<T>::from(v);
        if (xc) xc[(int64_t)i * ldx + f] = Elem<T>::from(vc);
        tile[r][tx] = vc;
        if (rowsq_part) {
            float s = wave_sum(v * v);
            if (tx == 0) sq[ty][r] = s;   // each r is handled by exactly one ty
        }
    }
    __syncthreads();
    if (xct) {
#pragma unroll 4
        for (int r = ty; r < 64; r += 4)   // r indexes f within the tile, tx indexes i
            xct[(int64_t)(f0 + r) * ldt + i0 + tx] = Elem<T>::from(tile[tx][r]);
    }
    if (rowsq_part && threadIdx.x < 64) {
        const int r = threadIdx.x;
        rowsq_part[(int64_t)blockIdx.y * gridDim.x * 64 + i0 + r] = sq[r & 3][r];
    }
}

__global__ void rowsq_reduce_kernel(const float* __restrict__ part, int nparts, int Bp, float* __restrict__ rowsq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp) return;
    float t = 0.f;
    for (int p = 0; p < nparts; ++p) t += part[(int64_t)p * Bp + i];
    rowsq[i] = t;
}

}  // namespace dae

using namespace dae;

int dae::launch_gather_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx, int B, int F,
                           int dtype, void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq, int corr_mode,
                           const uint32_t* keep_bits, uint64_t seed, uint32_t rng_stream, float corr_frac, float scale,
                           uint32_t* xc_bits, int64_t ldw, const LabelJob* label_job, hipStream_t st, uint32_t* x_bits) {
    DAE_CHECK_ARG(indptr && indices && row_idx, "gather_csr: null CSR / row_idx");
    DAE_CHECK_ARG(B > 0 && F > 0, "gather_csr: B=%d F=%d", B, F);
    DAE_CHECK_ARG(ldx >= F && ldx % DAE_PAD == 0, "gather_csr: ldx=%lld must be the padded feature count", (long long)ldx);
    DAE_CHECK_ARG(dtype == DAE_BF16 || dtype == DAE_F32, "gather_csr: bad dtype");
    DAE_CHECK_ARG(corr_mode != DAE_CORR_KEEPBITS || keep_bits, "gather_csr: keep_bits is null");
    DAE_CHECK_ARG(!xct || ldt >= dae_pad(B), "gather_csr: ldt too small");
    DAE_CHECK_ARG(!xc_bits || (!values && scale == 1.0f), "gather_csr: the bit-packed x~ needs binary data (values == NULL) and scale == 1");
    DAE_CHECK_ARG(!xc_bits || ldw >= ldx / 32, "gather_csr: ldw too small");
    DAE_CHECK_ARG(!x_bits || (!values && ldw >= ldx / 32), "gather_csr: the bit image of x needs binary data (values == NULL) and ldw >= Fp/32");
    DAE_CHECK_ARG(!label_job || label_job->Bp <= 1024, "gather_csr: the in-kernel label statistics need a padded batch <= 1024");
    const int Bp = (int)dae_pad(B);
    const unsigned chunks = (unsigned)((ldx + GATHER_CW - 1) / GATHER_CW);
    dim3 grid(Bp, chunks + (label_job ? 1u : 0u)), block(GATHER_THREADS);
    LabelJob job; memset(&job, 0, sizeof(job));
    if (label_job) job = *label_job;
    const int label_slice = label_job ? (int)chunks : -1;
    if (dtype == DAE_BF16)
        hipLaunchKernelGGL((gather_csr_kernel<bf16_t>), grid, block, 0, st, indptr, indices, values, row_idx, B, F,
                           (bf16_t*)x, (bf16_t*)xc, ldx, (bf16_t*)xct, ldt, rowsq, corr_mode, keep_bits, seed, rng_stream,
                           corr_frac, scale, xc_bits, ldw, job, label_slice, x_bits);
    else
        hipLaunchKernelGGL((gather_csr_kernel<float>), grid, block, 0, st, indptr, indices, values, row_idx, B, F,
                           (float*)x, (float*)xc, ldx, (float*)xct, ldt, rowsq, corr_mode, keep_bits, seed, rng_stream,
                           corr_frac, scale, xc_bits, ldw, job, label_slice, x_bits);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_gather_csr_bits(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx,
                                   int32_t B, int32_t F, int32_t dtype, void* x, void* xc, int64_t ldx, void* xct, int64_t ldt,
                                   float* rowsq, int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed,
                                   uint32_t rng_stream, float corr_frac, float scale, uint32_t* xc_bits, int64_t ldw,
                                   void* stream) {
    return launch_gather_csr(indptr, indices, values, row_idx, B, F, dtype, x, xc, ldx, xct, ldt, rowsq, corr_mode, keep_bits, seed,
                             rng_stream, corr_frac, scale, xc_bits, ldw, nullptr, (hipStream_t)stream);
}

extern "C" int dae_gather_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx,
                              int32_t B, int32_t F, int32_t dtype, void* x, void* xc, int64_t ldx, void* xct, int64_t ldt,
                              float* rowsq, int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed,
                              uint32_t rng_stream, float corr_frac, float scale, void* stream) {
    return dae_gather_csr_bits(indptr, indices, values, row_idx, B, F, dtype, x, xc, ldx, xct, ldt, rowsq, corr_mode, keep_bits,
                               seed, rng_stream, corr_frac, scale, nullptr, 0, stream);
}

extern "C" int dae_gather_dense(const float* data, int64_t ld_data, const int32_t* row_idx, int32_t B, int32_t F,
                                int32_t dtype, void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq,
                                float* rowsq_scratch, int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed,
                                uint32_t rng_stream, float corr_frac, float scale, void* stream) {
    DAE_CHECK_ARG(data && row_idx, "gather_dense: null input");
    DAE_CHECK_ARG(B > 0 && F > 0 && ld_data >= F, "gather_dense: bad shape");
    DAE_CHECK_ARG(ldx >= F && ldx % DAE_PAD == 0, "gather_dense: ldx must be the padded feature count");
    DAE_CHECK_ARG(dtype == DAE_BF16 || dtype == DAE_F32, "gather_dense: bad dtype");
    DAE_CHECK_ARG(corr_mode != DAE_CORR_KEEPBITS || keep_bits, "gather_dense: keep_bits is null");
    DAE_CHECK_ARG(!rowsq || rowsq_scratch, "gather_dense: rowsq needs rowsq_scratch[(Fp/64) x Bp]");
    const int Bp = (int)dae_pad(B);
    dim3 grid(Bp / 64, (unsigned)(ldx / 64)), block(256);
    hipStream_t st = (hipStream_t)stream;
    float* part = rowsq ? rowsq_scratch : nullptr;
    if (dtype == DAE_BF16)
        hipLaunchKernelGGL((gather_dense_kernel<bf16_t>), grid, block, 0, st, data, ld_data, row_idx, B, F, (bf16_t*)x,
                           (bf16_t*)xc, ldx, (bf16_t*)xct, ldt, part, corr_mode, keep_bits, seed, rng_stream, corr_frac, scale);
    else
        hipLaunchKernelGGL((gather_dense_kernel<float>), grid, block, 0, st, data, ld_data, row_idx, B, F, (float*)x,
                           (float*)xc, ldx, (float*)xct, ldt, part, corr_mode, keep_bits, seed, rng_stream, corr_frac, scale);
    DAE_CHECK_LAUNCH();
    if (rowsq) {
        hipLaunchKernelGGL(rowsq_reduce_kernel, dim3((Bp + 255) / 256), dim3(256), 0, st, part, (int)(ldx / 64), Bp, rowsq);
        DAE_CHECK_LAUNCH();
    }
    return 0;
}
