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
    LabelJob job, int label_slice, uint32_t* __restrict__ x_bits, T* __restrict__ x2) {
    // x2 (split-bf16 mode, valued data): lo image of the clean rows, x = x + x2 to 2^-17; it uses the x~ tile, so xc must be NULL then
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
        if (xc || x2) *reinterpret_cast<i32x4*>(&lxc[k]) = i32x4{0, 0, 0, 0};
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
                if (x2) lxc[col - c0] = elem_residual<T>(v);
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
        if (x2) *reinterpret_cast<i32x4*>(&x2[(int64_t)i * ldx + c0 + k]) = *reinterpret_cast<const i32x4*>(&lxc[k]);
    }
    // bit-packed x~ (binary inputs): bit b of word w of row i <=> feature 32*w + b is kept.  Operand of gemm_encode_bits.
    if (xc_bits && tid < (ncol >> 5)) xc_bits[(int64_t)i * ldw + (c0 >> 5) + tid] = lbits[tid];
    // bit image of the CLEAN row (binary inputs): what the decode epilogue reads instead of the dense x tile
    if (x_bits && tid < (ncol >> 5)) x_bits[(int64_t)i * ldw + (c0 >> 5) + tid] = lbits_x[tid];
}

// four / two consecutive elements as ONE store (the destinations are 8- resp. 16-byte aligned: padded leading dimensions,
// column offsets that are multiples of 4 / 2)
__device__ __forceinline__ void store4(bf16_t* dst, const float* v) {
    uint2 u;
    u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(dst) = u;
}
__device__ __forceinline__ void store4(float* dst, const float* v) { *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store2(bf16_t* dst, float a, float b) {
    *reinterpret_cast<uint32_t*>(dst) = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
}
__device__ __forceinline__ void store2(float* dst, float a, float b) { *reinterpret_cast<float2*>(dst) = make_float2(a, b); }

// Dense ndarray input (autoencoder.py:143 sparse_input False; dense masking utils.py:107-109).
// 64 x 64 tiles.  A thread owns four neighbouring features of a row: one 16-byte read of the fp32 row, ONE Philox evaluation for
// the four keep decisions (the per-element draw made this kernel VALU-bound: 10 rounds x 40 M elements at F = 50000), 8-byte
// stores of the bf16 x / x~ rows; the x~^T operand goes through an LDS transpose and is written as packed pairs.
// VEC = false is the same arithmetic element by element (unaligned rows, F not a multiple of 4).
template <typename T, bool VEC, int TR, int TC>
__global__ __launch_bounds__(256) void gather_dense_kernel(
    const float* __restrict__ data, int64_t ld_data, const int32_t* __restrict__ row_idx, int B, int F,
    T* __restrict__ x, T* __restrict__ xc, int64_t ldx, T* __restrict__ xct, int64_t ldt, float* __restrict__ rowsq_part,
    int corr_mode, const uint32_t* __restrict__ keep_bits, uint64_t seed, uint32_t stream, float corr_frac, float scale, int res) {
    // res != 0 (split-bf16 mode, second launch): every image receives the RESIDUAL v - bf16(v) instead of v (the lo parts of x, x~, x~^T)
    // TR x TC tile: the fp32 rows are read in runs of 4 TC bytes, x~ leaves in runs of TC elements, x~^T in runs of TR elements
    extern __shared__ __attribute__((aligned(16))) float gd_smem[];         // (dynamic: the 128 x 128 tile is 66 KiB)
    float (*tile)[TC + 1] = reinterpret_cast<float (*)[TC + 1]>(gd_smem);
    float* sq = gd_smem + TR * (TC + 1);
    constexpr int TPR = TC / 4, RPP = 256 / TPR;                     // threads per row (4 features each), rows per pass
    static_assert(TPR == 16 || TPR == 32 || TPR == 64, "a row's threads must sit in one wave");
    const int i0 = blockIdx.x * TR, f0 = blockIdx.y * TC;
    const int c4 = threadIdx.x % TPR, rr = threadIdx.x / TPR;
#pragma unroll
    for (int k = 0; k < TR / RPP; ++k) {
        const int r = rr + RPP * k, i = i0 + r, f = f0 + 4 * c4;
        float v[4] = {0.f, 0.f, 0.f, 0.f}, vc[4] = {0.f, 0.f, 0.f, 0.f};
        if (i < B && f < F) {
            const int64_t row = row_idx[i];
            if (VEC) {
                const float4 q = *reinterpret_cast<const float4*>(data + row * ld_data + f);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (f + j < F) ? data[row * ld_data + f + j] : 0.f;
            }
            bool keep[4] = {true, true, true, true};
            if (corr_mode == DAE_CORR_KEEPBITS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint64_t e = (uint64_t)row * (uint64_t)F + (uint64_t)(f + j);      // C order of the ndarray (utils.py:108)
                    keep[j] = (f + j < F) && ((keep_bits[e >> 5] >> (e & 31)) & 1u);
                }
            } else if (corr_mode == DAE_CORR_PHILOX_MASK) {
                const uint4 o = philox_dense4((uint32_t)row, (uint32_t)(f >> 2), seed, stream);
                keep[0] = philox_word_uniform(o.x) >= corr_frac; keep[1] = philox_word_uniform(o.y) >= corr_frac;
                keep[2] = philox_word_uniform(o.z) >= corr_frac; keep[3] = philox_word_uniform(o.w) >= corr_frac;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) vc[j] = keep[j] ? v[j] * scale : 0.f;
            if (res) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] -= bf2f(f2bf(v[j])); vc[j] -= bf2f(f2bf(vc[j])); }
            }
        }
        if (f < ldx) {                                                  // (the last tile of a 128-padded row may hang over at TC = 256)
            if (x) store4(x + (int64_t)i * ldx + f, v);
            if (xc) store4(xc + (int64_t)i * ldx + f, vc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[r][4 * c4 + j] = vc[j];
        if (rowsq_part) {
            float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            s2 = row16_sum(s2);                                                            // 16 lanes of a row sit in one DPP row
            if (TPR >= 32) s2 += __shfl_xor(s2, 16);
            if (TPR >= 64) s2 += __shfl_xor(s2, 32);
            if (c4 == 0) sq[r] = s2;
        }
    }
    __syncthreads();
    if (xct) {
        constexpr int LPF = TR / 2, FPP = 256 / LPF;                 // lanes per feature row (packed pairs of consecutive i), feature rows per pass
        const int l2 = threadIdx.x % LPF, fr0 = threadIdx.x / LPF;
#pragma unroll
        for (int p = 0; p < TC / FPP; ++p) {
            const int fr = fr0 + FPP * p;
            if (f0 + fr < ldx) store2(xct + (int64_t)(f0 + fr) * ldt + i0 + 2 * l2, tile[2 * l2][fr], tile[2 * l2 + 1][fr]);
        }
    }
    if (rowsq_part && threadIdx.x < TR) rowsq_part[(int64_t)blockIdx.y * gridDim.x * TR + i0 + threadIdx.x] = sq[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// Salt-and-pepper corruption of one mini-batch on the device (utils.salt_and_pepper_noise, utils.py:118-144).
// One workgroup per batch row.  An LDS table of F entries records, per column, the LAST draw that hit it (atomicMax of
// t*2 + coin, t = draw index: "later writes win", as the reference's sequential loop); the row is then rebuilt in column
// order: overridden columns take lo / hi, the others keep their stored value, zeros are dropped, and the ordered compaction
// (ballot ranks per 256-column strip) writes a sorted CSR row.
// ------------------------------------------------------------------------------------------------
constexpr int SP_THREADS = 256;

__global__ __launch_bounds__(SP_THREADS) void salt_pepper_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                                 const float* __restrict__ values, const int32_t* __restrict__ row_idx, int F,
                                                                 int v, float lo, float hi, uint64_t seed, uint32_t stream,
                                                                 int64_t* __restrict__ out_span, int32_t* __restrict__ out_indices,
                                                                 float* __restrict__ out_values, int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* hit = reinterpret_cast<uint32_t*>(smem);               // [F]  0 = untouched, else (t + 1) * 2 + coin of the last draw
    float* val = reinterpret_cast<float*>(smem + (size_t)F * 4);     // [F]  stored value of the clean row (0 where nothing is stored)
    __shared__ int wave_cnt[SP_THREADS / 64];
    __shared__ int base_s;
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = row_idx[i];
    for (int k = tid; k < F; k += SP_THREADS) { hit[k] = 0u; val[k] = 0.f; }
    if (tid == 0) base_s = 0;
    __syncthreads();
    const int64_t s0 = indptr[row], e0 = indptr[row + 1];
    for (int64_t k = s0 + tid; k < e0; k += SP_THREADS) {
        const int c = indices[k];
        if (c < F) val[c] = values ? values[k] : 1.0f;
    }
    for (int t = tid; t < v; t += SP_THREADS) {
        const uint4 o = philox4x32_10(make_uint4((uint32_t)row, (uint32_t)t, stream, 1u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const uint32_t c = __umulhi(o.x, (uint32_t)F);               // floor(u * F), u = o.x / 2^32
        const uint32_t coin = o.y >> 31;                             // 0: u2 < 0.5 -> lo, 1: hi
        atomicMax(&hit[c], (uint32_t)(t + 1) * 2u + coin);
    }
    __syncthreads();
    const int64_t obase = (int64_t)i * cap;
    for (int c0 = 0; c0 < F; c0 += SP_THREADS) {
        const int c = c0 + tid;
        float x = 0.f;
        if (c < F) {
            const uint32_t h = hit[c];
            x = h ? ((h & 1u) ? hi : lo) : val[c];
        }
        const bool nz = x != 0.f;
        const unsigned long long m = __ballot(nz);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int before = base_s;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w];
        if (nz) {
            const int o = before + __popcll(m & ((1ull << lane) - 1ull));
            if (o < cap) { out_indices[obase + o] = c; out_values[obase + o] = x; }
        }
        __syncthreads();
        if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) { out_span[2 * i] = obase; out_span[2 * i + 1] = obase + min(base_s, cap); }
}

// ------------------------------------------------------------------------------------------------
// Fused corrupt + gather + encode for CSR inputs -- what the reference's graph does in ONE op,
// tf.sparse.matmul(x~, W) (autoencoder.py:377,389), followed by + b_h, the activation and - act(b_h):
//     h[i, :] = act( sum_{e in row(i), kept(e)} scale * v_e * W[col_e, :] + b_h ) - act(b_h)
// The dense [B x F] image of x~ is never formed: a batch row holds ~200 of 10^4 features, so the kernel sums ~140 rows
// of W per batch row (0.11 GFLOP instead of the 8 GFLOP of the dense contraction) and is bound by the L2 -> VGPR rate of
// those W-row reads, not by MFMA.  Decomposition for the L2: the H columns are cut into slices of 128 (256 bytes of a bf16 W
// row); block b works on slice b % n_slices, and the dispatcher places block b on XCD b % 8, so with 4 slices every XCD
// re-reads only ITS 2.6 MB column slice of W_lo from its private 4 MiB L2 (placement affects speed only).
// Work split: one workgroup = 8 batch rows x one slice, one row per wave (448 workgroups at B = 800, H = 500: one resident
// round on 256 CUs).  A lane owns stored entries of the row (coalesced id / value / keep-decision reads, 256 entries per pass);
// the kept ones are compacted into an LDS list and walked 4 at a time: lane (sub, part) loads 16 bytes (part) of the W-row
// slice of entry `sub`, so one wave instruction fetches 4 W-row slices (1 KiB) and 18 of them are in flight per lane.  Per-lane
// fp32 accumulators (8 columns), one butterfly over the 4 entry groups at the end of the row, fixed order -> deterministic.  Epilogue: bias, activation, and every image of h the step needs (fp32, low precision, h^T
// via an LDS transpose, split-bf16 Gram operands).
// The workgroups of a row group also produce the batch's side images, one task per slice: the bit image of the CLEAN rows
// (decode epilogue), the scatter of kept entries into x~^T (dW GEMM operand), sum of squares (cosine_proximity).
// ------------------------------------------------------------------------------------------------
struct EncCsrArgs {
    const int64_t* indptr; const int32_t* indices; const float* values; const int32_t* row_idx;
    int B, Bp, F, H, Hp;
    const void* W; int64_t ldw;                 // [Fp x ldw] row-major, element type WT
    const float* bh;
    int corr_mode; const uint32_t* keep_bits; uint64_t seed; uint32_t stream; float corr_frac, scale;
    int enc_act;
    float* h_f32; void* h_lo; int64_t ldh; void* h_t; int64_t ldht; bf16_t* hcat_a; bf16_t* hcat_b;
    void* h_t2;                                 // lo image of h^T (split-bf16 mode: h = hi + lo, h_t holds hi) or NULL
    uint32_t* x_bits; int64_t ldxb;             // clean bit image [Bp x ldxb] (binary data) or NULL
    void* xct; int64_t ldt;                     // x~^T [Fp x ldt] scatter target (pre-zeroed) or NULL
    void* xct2;                                 // split-bf16 mode, x~ not exact in bf16: lo image of x~^T (same layout, pre-zeroed) or NULL
    int xct_rm;                                 // 1: xct / xct2 are row-major x~ [Bp x ldt] (entry (i, col) at i * ldt + col)
    uint32_t* xtb; int64_t ldxt;                // x~^T as a BIT image [Fp x ldxt words] (pre-zeroed; bit i of row f <=> entry (i, f) kept) or NULL
    int xtl_off, Fp;                            // xtl_off > 0: LDS byte image [Fp] of the workgroup's 8 batch rows at that offset (else global atomics)
    float* rowsq;                               // [Bp] or NULL
    int n_slices;
    LabelJob job; int label_block;
};

constexpr int ENC_ROWS = 8;                     // batch rows per workgroup: one per wave (8 waves = 512 threads)
constexpr int ENC_THREADS = 64 * ENC_ROWS;

// One W-row slice of COLS columns is read by the 16 lanes of an entry group, CPL = COLS / 16 columns per lane.
//   bf16 shadow, 128 columns (256 B): 16 B per lane      fp32 master, 128 columns (512 B): 32 B per lane
//   fp32 master,  64 columns (256 B): 16 B per lane  -- 8 slices of 2.6 MB at 10000 x 500, one per XCD L2, where the 128-column
//   fp32 slice (5.2 MB) would not fit a 4 MiB L2
template <typename WT, int CPL> struct WRow;
template <> struct WRow<bf16_t, 8> {
    typedef i32x4 Raw;
    static constexpr int BATCH = 18;              // W-row loads in flight per lane (4 VGPRs each)
    static __device__ __forceinline__ Raw load(const char* p) { return *reinterpret_cast<const i32x4*>(p); }
    static __device__ __forceinline__ void fma(const Raw& v, float w, float (&acc)[8]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // a packed pair of the library's 16-bit storage format (bf16: the two halves ARE the fp32 high words; fp16 build: two v_cvt_f32_f16)
            if constexpr (kF16) {
                acc[2 * q] = fmaf(w, bf2f((bf16_t)((uint32_t)v[q] & 0xffffu)), acc[2 * q]);
                acc[2 * q + 1] = fmaf(w, bf2f((bf16_t)((uint32_t)v[q] >> 16)), acc[2 * q + 1]);
            } else {
                acc[2 * q] = fmaf(w, __uint_as_float(((uint32_t)v[q]) << 16), acc[2 * q]);
                acc[2 * q + 1] = fmaf(w, __uint_as_float(((uint32_t)v[q]) & 0xffff0000u), acc[2 * q + 1]);
            }
        }
    }
};
template <> struct WRow<float, 8> {
    struct Raw { f32x4 a, b; };
    static constexpr int BATCH = 9;               // 8 VGPRs each
    static __device__ __forceinline__ Raw load(const char* p) {
        Raw r; r.a = *reinterpret_cast<const f32x4*>(p); r.b = *reinterpret_cast<const f32x4*>(p + 16); return r;
    }
    static __device__ __forceinline__ void fma(const Raw& v, float w, float (&acc)[8]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q] = fmaf(w, v.a[q], acc[q]); acc[4 + q] = fmaf(w, v.b[q], acc[4 + q]); }
    }
};
template <> struct WRow<float, 4> {
    typedef f32x4 Raw;
    static constexpr int BATCH = 18;
    static __device__ __forceinline__ Raw load(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void fma(const Raw& v, float w, float (&acc)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fmaf(w, v[q], acc[q]);
    }
};

// LDS carve of the kernel (bytes): pre-activation tile, transposed low-precision tile, per-wave kept-entry lists, clean bit rows
__host__ __device__ constexpr size_t enc_lds_fixed(int cols, int tsize) {
    return (size_t)ENC_ROWS * cols * 4 + (size_t)cols * ENC_ROWS * tsize + (size_t)ENC_ROWS * 256 * 8;
}

// WT: element type of the weight image that is read (bf16 shadow, or the fp32 master);  T: element type of the activation
// images that are written (h_lo, h^T, x~^T);  COLS: H columns per workgroup ("slice")
template <typename WT, typename T, int COLS>
__global__ __launch_bounds__(ENC_THREADS, 4) void encode_csr_kernel(EncCsrArgs a) {
    constexpr int CPL = COLS / 16;                 // columns per lane of an entry group
    constexpr int EPL = COLS / 64;                 // columns per lane of the epilogue (thread = row x EPL columns)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x == a.label_block) { label_stats_block<ENC_THREADS>(a.job, smem); return; }
    float* zt = reinterpret_cast<float*>(smem);                                           // [ENC_ROWS][COLS] pre-activations of this slice
    T* ht = reinterpret_cast<T*>(smem + ENC_ROWS * COLS * 4);                             // [COLS][ENC_ROWS] transposed low-precision h
    int2* const lists = reinterpret_cast<int2*>(smem + ENC_ROWS * COLS * 4 + COLS * ENC_ROWS * sizeof(T));   // [ENC_ROWS][256] kept (column, value)
    uint32_t* xb = reinterpret_cast<uint32_t*>(smem + enc_lds_fixed(COLS, (int)sizeof(T)));                  // [ENC_ROWS][ldxb] clean bit rows
    const int slice = blockIdx.x % a.n_slices, i0 = (blockIdx.x / a.n_slices) * ENC_ROWS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6), i = i0 + r;  // this wave's batch row
    const int sub = lane >> 4, part = lane & 15;                         // 4 entries per load instruction, 16 lanes per W-row slice
    // side images: task t is produced by the workgroups of slice t % n_slices
    const bool do_xbits = a.x_bits && slice == 0;
    const bool do_xct = (a.xct || a.xtb) && slice == 1 % a.n_slices;
    const bool do_rowsq = a.rowsq && slice == 2 % a.n_slices;
    // x~^T bits of this workgroup's 8 batch rows (i0 is a multiple of 8): ONE byte per feature, assembled in LDS (bit r of byte f =
    // row i0 + r keeps feature f) and stored as bytes -- the workgroup owns byte i0 / 8 of every feature row, so no global atomics
    uint32_t* xtl = reinterpret_cast<uint32_t*>(smem + a.xtl_off);
    const bool xt_lds = do_xct && a.xtb && a.xtl_off > 0;
    if (do_xbits) for (int k = tid; k < ENC_ROWS * (int)a.ldxb; k += ENC_THREADS) xb[k] = 0u;
    if (xt_lds) for (int k = tid; k < a.Fp / 4; k += ENC_THREADS) xtl[k] = 0u;
    if (do_xbits || xt_lds) __syncthreads();
    const char* Wb = reinterpret_cast<const char*>(a.W) + (int64_t)slice * COLS * sizeof(WT) + part * (CPL * sizeof(WT));
    const uint64_t ldw_b = (uint64_t)(a.ldw * (int64_t)sizeof(WT));
    T* xct = reinterpret_cast<T*>(a.xct);
    int2* const mylist = lists + r * 256;
    float acc[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) acc[q] = 0.f;
    float sq = 0.f;
#ifdef DAE_ENC_PROBE
    const bool probe_skip_rows = (DAE_ENC_PROBE & 2) != 0;              // probe: no entry phase at all
#else
    const bool probe_skip_rows = false;
#endif
    if (i < a.B && !probe_skip_rows) {
        const int64_t row = a.row_idx[i];
        const int64_t s0 = a.indptr[row], e0 = a.indptr[row + 1];
        for (int64_t base = s0; base < e0; base += 256) {
            // a lane owns 4 stored entries of the pass.  All reads (ids, values, keep words) are issued UNCONDITIONALLY on clamped
            // addresses, group by group, so that they are in flight together: a per-entry branch around a load makes hipcc wait
            // vmcnt(0) per entry, i.e. four dependent L2 round trips instead of one.
            int col[4]; float vv[4];
            int64_t kc[4];
            uint32_t kw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                kc[u] = min(base + u * 64 + lane, e0 - 1);
                col[u] = a.indices[kc[u]];
            }
            if (a.values) {
#pragma unroll
                for (int u = 0; u < 4; ++u) vv[u] = a.values[kc[u]];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) vv[u] = 1.0f;
            }
            if (a.corr_mode == DAE_CORR_KEEPBITS) {
#pragma unroll
                for (int u = 0; u < 4; ++u) kw[u] = a.keep_bits[kc[u] >> 5];
            }
            // keep decisions, side images, and compaction of the KEPT entries into this wave's LDS list (ballot ranks: storage
            // order is preserved, so the summation order -- and the result -- is deterministic); dropped and padding entries
            // cost no W-row read
            int nkept = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool valid = base + u * 64 + lane < e0 && col[u] < a.F;
                bool keep = valid;
                if (a.corr_mode == DAE_CORR_KEEPBITS) keep = valid && ((kw[u] >> (kc[u] & 31)) & 1u);
                else if (a.corr_mode == DAE_CORR_PHILOX_MASK) keep = valid && philox_uniform((uint64_t)kc[u], a.seed, a.stream) >= a.corr_frac;
                const float v = valid ? vv[u] : 0.f;
                const float w = keep ? v * a.scale : 0.f;
                if (do_xbits && valid) atomicOr(&xb[r * a.ldxb + (col[u] >> 5)], 1u << (col[u] & 31));
                if (do_xct && keep) {
                    // x~^T for the dW GEMM: a bit per kept entry (binary data; integer OR -> order-independent), or the dense scatter
                    if (xt_lds) atomicOr(&xtl[col[u] >> 2], 1u << (8 * (col[u] & 3) + r));
                    else if (a.xtb) atomicOr(&a.xtb[(int64_t)col[u] * a.ldxt + (i >> 5)], 1u << (i & 31));
                    else {
                        const int64_t o = a.xct_rm ? (int64_t)i * a.ldt + col[u] : (int64_t)col[u] * a.ldt + i;
                        xct[o] = Elem<T>::from(w);
                        if (a.xct2) reinterpret_cast<T*>(a.xct2)[o] = elem_residual<T>(w);
                    }
                }
                if (do_rowsq) sq += v * v;
                const bool kp = w != 0.f;
                const unsigned long long m = __ballot(kp);
                if (kp) mylist[nkept + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(col[u], __float_as_int(w));
                nkept += __popcll(m);
            }
#ifdef DAE_ENC_PROBE
            if (DAE_ENC_PROBE & 1) nkept = 0;                            // probe: no W-row reads
#endif
            // walk the list 4 entries at a time (one per 16-lane group): lane (sub, part) reads its CPL columns of the W-row slice
            // of entry `sub`; BATCH loads are issued back to back before the first use
            constexpr int NB = WRow<WT, CPL>::BATCH;
            for (int t0 = 0; t0 * 4 < nkept; t0 += NB) {
                typename WRow<WT, CPL>::Raw wr[NB];
                float wj[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int e = (t0 + j) * 4 + sub;
                    const int2 cw = mylist[min(e, 255)];
                    const bool on = e < nkept;
                    wj[j] = on ? __int_as_float(cw.y) : 0.f;
                    wr[j] = WRow<WT, CPL>::load(Wb + (uint64_t)(uint32_t)(on ? cw.x : 0) * ldw_b);
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) WRow<WT, CPL>::fma(wr[j], wj[j], acc);
            }
        }
    }
    // butterfly over the 4 entry groups (lane bits 4, 5); lanes 0..15 (sub == 0) end up with the row's sums
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        float t = acc[q];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        acc[q] = t;
    }
    if (sub == 0) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) zt[r * COLS + part * CPL + q] = acc[q];
    }
    if (do_rowsq) {
        sq = wave_sum(sq);
        if (lane == 0) a.rowsq[i] = (i < a.B) ? sq : 0.f;
    }
    __syncthreads();
    // ---- epilogue on the [8 rows x COLS columns] tile: thread = (row, EPL neighbouring columns) ----
#ifdef DAE_ENC_PROBE
    if (DAE_ENC_PROBE & 4) return;                                       // probe: no epilogue
#endif
    float hv[EPL];
    {
        const int col0 = slice * COLS + lane * EPL;
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int cl = lane * EPL + u, col = col0 + u;
            const float b = a.bh[col];
            const float z = zt[r * COLS + cl] + b;
            hv[u] = (i < a.B && col < a.H) ? act_apply(a.enc_act, z) - act_apply(a.enc_act, b) : 0.f;
            ht[cl * ENC_ROWS + r] = Elem<T>::from(hv[u]);
        }
        if (a.h_f32) {
            float* hp = a.h_f32 + (int64_t)i * a.ldh + col0;
            if constexpr (EPL == 2) *reinterpret_cast<float2*>(hp) = make_float2(hv[0], hv[1]);
            else hp[0] = hv[0];
        }
        if (a.h_lo) {
            T* hl = reinterpret_cast<T*>(a.h_lo) + (int64_t)i * a.ldh + col0;
            if constexpr (sizeof(T) == 2 && EPL == 2) *reinterpret_cast<uint32_t*>(hl) = (uint32_t)f2bf(hv[0]) | ((uint32_t)f2bf(hv[1]) << 16);
            else {
#pragma unroll
                for (int u = 0; u < EPL; ++u) hl[u] = Elem<T>::from(hv[u]);
            }
        }
        if (a.hcat_a) {   // split-bf16 operands of the Gram matrix: h = hi + lo, D ~= hi.hi + hi.lo + lo.hi
            bf16_t* pa = a.hcat_a + (int64_t)i * (3 * a.Hp) + col0;
            bf16_t* pb = a.hcat_b + (int64_t)i * (3 * a.Hp) + col0;
            if constexpr (EPL == 2) {
                const bf16_t hi0 = f2bf(hv[0]), hi1 = f2bf(hv[1]);
                const bf16_t lo0 = f2bf(hv[0] - bf2f(hi0)), lo1 = f2bf(hv[1] - bf2f(hi1));
                const uint32_t hi = (uint32_t)hi0 | ((uint32_t)hi1 << 16), lo = (uint32_t)lo0 | ((uint32_t)lo1 << 16);
                uint32_t* qa = reinterpret_cast<uint32_t*>(pa);
                uint32_t* qb = reinterpret_cast<uint32_t*>(pb);
                qa[0] = hi; qa[a.Hp / 2] = hi; qa[a.Hp] = lo;
                qb[0] = hi; qb[a.Hp / 2] = lo; qb[a.Hp] = hi;
            } else {
                const bf16_t hi = f2bf(hv[0]), lo = f2bf(hv[0] - bf2f(hi));
                pa[0] = hi; pa[a.Hp] = hi; pa[2 * a.Hp] = lo;
                pb[0] = hi; pb[a.Hp] = lo; pb[2 * a.Hp] = hi;
            }
        }
    }
    __syncthreads();
    if (a.h_t && tid < COLS) {                       // h^T: 8 batch columns of one feature row = one 16-byte (bf16) / 32-byte store
        T* dst = reinterpret_cast<T*>(a.h_t) + (int64_t)(slice * COLS + tid) * a.ldht + i0;
        const T* src = ht + tid * ENC_ROWS;
        *reinterpret_cast<i32x4*>(dst) = *reinterpret_cast<const i32x4*>(src);
        if constexpr (sizeof(T) == 4) *reinterpret_cast<i32x4*>(dst + 4) = *reinterpret_cast<const i32x4*>(src + 4);
    }
    if constexpr (sizeof(T) == 2) {
        if (a.h_t2) {                                // split-bf16: the lo image of h^T through the same transposing tile
            __syncthreads();
#pragma unroll
            for (int u = 0; u < EPL; ++u) ht[(lane * EPL + u) * ENC_ROWS + r] = elem_residual<T>(hv[u]);
            __syncthreads();
            if (tid < COLS)
                *reinterpret_cast<i32x4*>(reinterpret_cast<T*>(a.h_t2) + (int64_t)(slice * COLS + tid) * a.ldht + i0) =
                    *reinterpret_cast<const i32x4*>(ht + tid * ENC_ROWS);
        }
    }
    if (do_xbits) {
        for (int k = tid; k < ENC_ROWS * (int)a.ldxb; k += ENC_THREADS) {
            const int rr = k / (int)a.ldxb, w = k % (int)a.ldxb;
            a.x_bits[(int64_t)(i0 + rr) * a.ldxb + w] = xb[k];
        }
    }
    if (xt_lds) {                                    // (the barriers above ordered every LDS atomic before these reads)
        uint8_t* xt8 = reinterpret_cast<uint8_t*>(a.xtb) + (i0 >> 3);
        const int64_t ld8 = a.ldxt * 4;
        for (int k = tid; k < a.Fp / 4; k += ENC_THREADS) {
            uint32_t w = xtl[k];
            while (w) {
                const int q = __builtin_ctz(w) >> 3;
                xt8[(int64_t)(4 * k + q) * ld8] = (uint8_t)(w >> (8 * q));
                w &= ~(0xffu << (8 * q));
            }
        }
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

static int g_gather_tile = 0;      // dense gather tile shape (plan option "gather_tile", process-wide)
void dae::set_gather_tile(int v) { g_gather_tile = v & 3; }

int dae::launch_gather_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx, int B, int F,
                           int dtype, void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq, int corr_mode,
                           const uint32_t* keep_bits, uint64_t seed, uint32_t rng_stream, float corr_frac, float scale,
                           uint32_t* xc_bits, int64_t ldw, const LabelJob* label_job, hipStream_t st, uint32_t* x_bits, void* x2) {
    DAE_CHECK_ARG(indptr && indices && row_idx, "gather_csr: null CSR / row_idx");
    DAE_CHECK_ARG(!x2 || (x && !xc && dtype == DAE_BF16), "gather_csr: the lo image of x needs x, no x~ tile and bf16");
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
        DAE_LAUNCH((gather_csr_kernel<bf16_t>), grid, block, 0, st, indptr, indices, values, row_idx, B, F,
                           (bf16_t*)x, (bf16_t*)xc, ldx, (bf16_t*)xct, ldt, rowsq, corr_mode, keep_bits, seed, rng_stream,
                           corr_frac, scale, xc_bits, ldw, job, label_slice, x_bits, (bf16_t*)x2);
    else
        DAE_LAUNCH((gather_csr_kernel<float>), grid, block, 0, st, indptr, indices, values, row_idx, B, F,
                           (float*)x, (float*)xc, ldx, (float*)xct, ldt, rowsq, corr_mode, keep_bits, seed, rng_stream,
                           corr_frac, scale, xc_bits, ldw, job, label_slice, x_bits, (float*)nullptr);
    DAE_CHECK_LAUNCH();
    return 0;
}

// dynamic LDS of the launch for `dtype` activations, weights read as fp32 (w_f32) or as the low-precision shadow, and
// ldxb words per clean bit row (0 = no bit rows); the step driver asks this before it lets the launch emit the clean bit image
static int enc_cols(int dtype, int w_f32, int w32_cols) { return (dtype == DAE_BF16 && w_f32 && w32_cols == 64) ? 64 : 128; }
size_t dae::encode_csr_lds_bytes(int dtype, int w_f32, int w32_cols, int64_t ldxb) {
    const int cols = enc_cols(dtype, w_f32, w32_cols);
    size_t lds = enc_lds_fixed(cols, dtype == DAE_BF16 ? 2 : 4) + (size_t)ENC_ROWS * (size_t)ldxb * 4;
    return lds;
}

int dae::launch_encode_csr(const EncCsrLaunch& q, hipStream_t st) {
    DAE_CHECK_ARG(q.indptr && q.indices && q.row_idx && q.W && q.bh, "encode_csr: null input");
    DAE_CHECK_ARG(q.B > 0 && q.F > 0 && q.H > 0, "encode_csr: bad shape");
    DAE_CHECK_ARG(q.dtype == DAE_BF16 || q.dtype == DAE_F32, "encode_csr: bad dtype");
    DAE_CHECK_ARG(q.corr_mode != DAE_CORR_KEEPBITS || q.keep_bits, "encode_csr: keep_bits is null");
    const int Bp = (int)dae_pad(q.B), Hp = (int)dae_pad(q.H);
    DAE_CHECK_ARG(q.ldw >= Hp && q.ldh >= Hp && (!q.h_t || q.ldht >= Bp), "encode_csr: leading dimensions too small");
    DAE_CHECK_ARG(!q.x_bits || (!q.values && q.ldxb >= dae_pad(q.F) / 32), "encode_csr: the bit image of x needs binary data and ldxb >= Fp/32");
    DAE_CHECK_ARG(!q.xct || q.ldt >= (q.xct_rm ? (int)dae_pad(q.F) : Bp), "encode_csr: ldt too small");
    DAE_CHECK_ARG(!q.xtb || (!q.values && q.ldxt >= Bp / 32 && !q.xct), "encode_csr: the bit image of x~^T needs binary data, ldxt >= Bp/32 and no dense x~^T");
    DAE_CHECK_ARG(!q.label_job || q.label_job->Bp <= 1024, "encode_csr: in-kernel label statistics need a padded batch <= 1024");
    DAE_CHECK_ARG((q.hcat_a == nullptr) == (q.hcat_b == nullptr), "encode_csr: hcat_a/hcat_b must be given together");
    EncCsrArgs a;
    memset(&a, 0, sizeof(a));
    a.indptr = q.indptr; a.indices = q.indices; a.values = q.values; a.row_idx = q.row_idx;
    a.B = q.B; a.Bp = Bp; a.F = q.F; a.H = q.H; a.Hp = Hp; a.W = q.W; a.ldw = q.ldw; a.bh = q.bh;
    a.corr_mode = q.corr_mode; a.keep_bits = q.keep_bits; a.seed = q.seed; a.stream = q.rng_stream; a.corr_frac = q.corr_frac; a.scale = q.scale;
    a.enc_act = q.enc_act; a.h_f32 = q.h_f32; a.h_lo = q.h_lo; a.ldh = q.ldh; a.h_t = q.h_t; a.ldht = q.ldht; a.h_t2 = q.h_t2;
    a.hcat_a = (bf16_t*)q.hcat_a; a.hcat_b = (bf16_t*)q.hcat_b; a.x_bits = q.x_bits; a.ldxb = q.ldxb; a.xct = q.xct; a.ldt = q.ldt;
    a.xtb = q.xtb; a.ldxt = q.ldxt; a.xct2 = q.xct2; a.xct_rm = q.xct_rm;
    DAE_CHECK_ARG(!q.xct2 || (q.xct && q.dtype == DAE_BF16), "encode_csr: the lo image of x~^T needs the dense x~^T and bf16");
    const int cols = enc_cols(q.dtype, q.w_f32, q.w32_cols);
    a.rowsq = q.rowsq; a.n_slices = Hp / cols;
    const int nblk = a.n_slices * (Bp / ENC_ROWS);
    a.label_block = q.label_job ? nblk : -1;
    if (q.label_job) a.job = *q.label_job;
    size_t lds = encode_csr_lds_bytes(q.dtype, q.w_f32, q.w32_cols, q.x_bits ? q.ldxb : 0);
    a.Fp = (int)dae_pad(q.F);
    if (q.xtb && lds + (size_t)a.Fp <= 64 * 1024) { a.xtl_off = (int)lds; lds += (size_t)a.Fp; }     // else: global atomics
    if (q.label_job && lds < (size_t)LABEL_SMEM_BYTES) lds = LABEL_SMEM_BYTES;
    DAE_CHECK_ARG(lds <= 64 * 1024, "encode_csr: %zu B of LDS for the bit rows of %d features (route the clean rows through dae_gather_csr)", lds, q.F);
    dim3 grid(nblk + (q.label_job ? 1 : 0)), block(ENC_THREADS);
    // weight image: the low-precision shadow W_lo of the activation type, or -- w_f32, bf16 activations -- the fp32 master
    if (q.dtype == DAE_BF16) {
        if (!q.w_f32) DAE_LAUNCH((encode_csr_kernel<bf16_t, bf16_t, 128>), grid, block, lds, st, a);
        else if (cols == 64) DAE_LAUNCH((encode_csr_kernel<float, bf16_t, 64>), grid, block, lds, st, a);
        else DAE_LAUNCH((encode_csr_kernel<float, bf16_t, 128>), grid, block, lds, st, a);
    } else {
        DAE_LAUNCH((encode_csr_kernel<float, float, 128>), grid, block, lds, st, a);
    }
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_encode_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx,
                              int32_t B, int32_t F, int32_t H, int32_t dtype, const void* W_lo, int64_t ldw, const float* bh,
                              int32_t enc_act, int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed, uint32_t rng_stream,
                              float corr_frac, float scale, float* h_f32, void* h_lo, int64_t ldh, void* h_t, int64_t ldht,
                              void* hcat_a, void* hcat_b, uint32_t* x_bits, int64_t ldxb, void* xct, int64_t ldt, float* rowsq,
                              void* stream) {
    EncCsrLaunch q;
    memset(&q, 0, sizeof(q));
    q.indptr = indptr; q.indices = indices; q.values = values; q.row_idx = row_idx; q.B = B; q.F = F; q.H = H; q.dtype = dtype;
    q.W = W_lo; q.ldw = ldw; q.bh = bh; q.enc_act = enc_act; q.corr_mode = corr_mode; q.keep_bits = keep_bits; q.seed = seed;
    q.rng_stream = rng_stream; q.corr_frac = corr_frac; q.scale = scale; q.h_f32 = h_f32; q.h_lo = h_lo; q.ldh = ldh; q.h_t = h_t;
    q.ldht = ldht; q.hcat_a = hcat_a; q.hcat_b = hcat_b; q.x_bits = x_bits; q.ldxb = ldxb; q.xct = xct; q.ldt = ldt; q.rowsq = rowsq;
    return launch_encode_csr(q, (hipStream_t)stream);
}

extern "C" int dae_salt_pepper_batch(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx,
                                     int32_t B, int32_t F, int32_t v, float lo, float hi, uint64_t seed, uint32_t rng_stream,
                                     int64_t* out_span, int32_t* out_indices, float* out_values, int32_t cap, void* stream) {
    DAE_CHECK_ARG(indptr && indices && row_idx && out_span && out_indices && out_values, "salt_pepper_batch: null argument");
    DAE_CHECK_ARG(B > 0 && F > 0 && v >= 0 && cap > 0, "salt_pepper_batch: bad sizes");
    const size_t lds = (size_t)F * 8;
    DAE_CHECK_ARG(lds <= 150 * 1024, "salt_pepper_batch: %d features need %zu B of LDS", F, lds);
    static int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(salt_pepper_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    DAE_CHECK_ARG(attr_rc == 0, "salt_pepper_batch: hipFuncSetAttribute failed");
    DAE_LAUNCH(salt_pepper_kernel, dim3(B), dim3(SP_THREADS), lds, (hipStream_t)stream, indptr, indices, values, row_idx, F, v, lo, hi,
                       seed, rng_stream, out_span, out_indices, out_values, cap);
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

int dae::launch_gather_dense(const float* data, int64_t ld_data, const int32_t* row_idx, int32_t B, int32_t F,
                                int32_t dtype, void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq,
                                float* rowsq_scratch, int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed,
                                uint32_t rng_stream, float corr_frac, float scale, void* stream, int res) {
    DAE_CHECK_ARG(data && row_idx, "gather_dense: null input");
    DAE_CHECK_ARG(B > 0 && F > 0 && ld_data >= F, "gather_dense: bad shape");
    DAE_CHECK_ARG(ldx >= F && ldx % DAE_PAD == 0, "gather_dense: ldx must be the padded feature count");
    DAE_CHECK_ARG(dtype == DAE_BF16 || dtype == DAE_F32, "gather_dense: bad dtype");
    DAE_CHECK_ARG(corr_mode != DAE_CORR_KEEPBITS || keep_bits, "gather_dense: keep_bits is null");
    DAE_CHECK_ARG(!res || (dtype == DAE_BF16 && !rowsq), "gather_dense: the residual pass writes bf16 lo images and no row squares");
    DAE_CHECK_ARG(!rowsq || rowsq_scratch, "gather_dense: rowsq needs rowsq_scratch[(Fp/64) x Bp]");
    const int Bp = (int)dae_pad(B);
    // tile shape: 64 x 64 (option gather_tile = 0), 64 x 128 (1), 128 x 64 (2), 128 x 128 (3) -- rows x features
    const int tr = (g_gather_tile & 2) ? 128 : 64, tc = (g_gather_tile & 1) ? 128 : 64;
    dim3 grid(Bp / tr, (unsigned)((ldx + tc - 1) / tc)), block(256);
    hipStream_t st = (hipStream_t)stream;
    float* part = rowsq ? rowsq_scratch : nullptr;
    // 16-byte row reads need F % 4 == 0 and 16-byte aligned rows
    const bool vec = (F % 4 == 0) && (ld_data % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15) == 0);
#define DAE_GD(TT, VV, TR_, TC_) do {                                                                                                            \
        constexpr int ldsb = (TR_ * (TC_ + 1) + TR_) * 4;                                                                                         \
        static int attr_rc = ldsb > 64 * 1024 ? (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gather_dense_kernel<TT, VV, TR_, TC_>),    \
                                                                         hipFuncAttributeMaxDynamicSharedMemorySize, ldsb) : 0;                   \
        DAE_CHECK_ARG(attr_rc == 0, "gather_dense: hipFuncSetAttribute failed");                                                                   \
        DAE_LAUNCH((gather_dense_kernel<TT, VV, TR_, TC_>), grid, block, ldsb, st, data, ld_data, row_idx, B, F, (TT*)x, (TT*)xc, ldx,    \
                           (TT*)xct, ldt, part, corr_mode, keep_bits, seed, rng_stream, corr_frac, scale, res);                                        \
    } while (0)
#define DAE_GD_T(TT, VV) do { switch (g_gather_tile & 3) { case 0: DAE_GD(TT, VV, 64, 64); break; case 1: DAE_GD(TT, VV, 64, 128); break; \
                                                           case 2: DAE_GD(TT, VV, 128, 64); break; default: DAE_GD(TT, VV, 128, 128); break; } } while (0)
    if (dtype == DAE_BF16) { if (vec) DAE_GD_T(bf16_t, true); else DAE_GD_T(bf16_t, false); }
    else { if (vec) DAE_GD_T(float, true); else DAE_GD_T(float, false); }
#undef DAE_GD_T
#undef DAE_GD
    DAE_CHECK_LAUNCH();
    if (rowsq) {
        DAE_LAUNCH(rowsq_reduce_kernel, dim3((Bp + 255) / 256), dim3(256), 0, st, part, (int)grid.y, Bp, rowsq);
        DAE_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int dae_gather_dense(const float* data, int64_t ld_data, const int32_t* row_idx, int32_t B, int32_t F,
                                int32_t dtype, void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq,
                                float* rowsq_scratch, int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed,
                                uint32_t rng_stream, float corr_frac, float scale, void* stream) {
    return launch_gather_dense(data, ld_data, row_idx, B, F, dtype, x, xc, ldx, xct, ldt, rowsq, rowsq_scratch, corr_mode, keep_bits, seed,
                               rng_stream, corr_frac, scale, stream, 0);
}
