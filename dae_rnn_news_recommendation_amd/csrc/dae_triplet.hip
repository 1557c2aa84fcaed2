// dae_triplet.hip -- online triplet miners on the dot-product Gram matrix D = h h^T.
//
// batch_all  (triplet_loss_utils.py:79-131): the reference materialises ~8 tensors of B^3 elements
//   (2 GB each at B=800).  Here one workgroup per anchor streams its D row into LDS, compacts the
//   positives (same label, != anchor) and negatives (different label) in index order, and sweeps the
//   |P| x |N| rectangle once per role with the operands in registers/LDS -- nothing B^3 ever exists.
//   Outputs per anchor: sum softplus(D[a,n]-D[a,p]), #positive triplets, and the un-normalised
//   gradient row G[a,:] (sum of sigmoids with the role's sign).  N_valid / data_weight are integer
//   closed forms of the label histogram (dae_label_stats).
// batch_hard (triplet_loss_utils.py:202-259): one workgroup per anchor, wavefront-shuffle max/min
//   reductions; reproduces the reference's quirks literally (invalid negatives contribute 0, invalid
//   positives are shifted by the row max, float-equality data_weight, gradient ties split equally,
//   gradient through the row-max shift).
#include "dae_common.h"

namespace dae {

constexpr int TRIP_THREADS = 256;
constexpr int TRIP_MAX_B = 4096;

// block-wide helpers (256 threads = 4 waves)
__device__ __forceinline__ float block_sum_f(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ unsigned block_sum_u(unsigned v, unsigned* red) {
    v = wave_sum_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_f(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_min_f(float v, float* red) {
    v = wave_min(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

// softplus(t) and sigmoid(t) from one exp (TF: -log_sigmoid(-t) = softplus(t); SoftplusGrad = sigmoid)
__device__ __forceinline__ void softplus_sigmoid(float t, float& sp, float& sg) {
    const float en = __expf(-fabsf(t));                       // in (0,1]
    const float r = __frcp_rn(1.0f + en);
    const float l = en < 1e-4f ? en * (1.0f - 0.5f * en) : __logf(1.0f + en);
    sp = fmaxf(t, 0.f) + l;
    sg = t >= 0.f ? r : en * r;
}
__device__ __forceinline__ float sigmoid_only(float t) {
    const float en = __expf(-fabsf(t));
    const float r = __frcp_rn(1.0f + en);
    return t >= 0.f ? r : en * r;
}

template <bool POS_ONLY>
__global__ __launch_bounds__(TRIP_THREADS) void batch_all_kernel(const float* __restrict__ D_slabs, int d_splits,
                                                                 int64_t slab_stride, int64_t ldd,
                                                                 const int32_t* __restrict__ labels, int B, int Bp,
                                                                 float* __restrict__ loss_part, uint32_t* __restrict__ npos_part,
                                                                 float* __restrict__ G, uint32_t* __restrict__ role_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* pu = reinterpret_cast<float*>(smem);              // positives' D[a,p]      [Bp]
    float* nv = pu + Bp;                                      // negatives' D[a,n]      [Bp]
    int* pidx = reinterpret_cast<int*>(nv + Bp);              // [Bp]
    int* nidx = pidx + Bp;                                    // [Bp]
    int* scan = nidx + Bp;                                    // [2][TRIP_THREADS + 1]
    float* red = reinterpret_cast<float*>(scan + 2 * (TRIP_THREADS + 1));   // [8]
    unsigned* redu = reinterpret_cast<unsigned*>(red + 4);

    const int a = blockIdx.x;
    const int tid = threadIdx.x;
    const int32_t la = labels[a];
    float* Grow = G + (int64_t)a * Bp;
    uint32_t* Rrow = POS_ONLY ? role_cnt + (int64_t)a * Bp : nullptr;

    // zero the gradient row (covers j == a, j >= B)
    for (int j = tid; j < Bp; j += TRIP_THREADS) { Grow[j] = 0.f; if (POS_ONLY) Rrow[j] = 0u; }

    // ---- deterministic compaction: thread t owns the contiguous index range [t*C, (t+1)*C) ----
    const int C = (B + TRIP_THREADS - 1) / TRIP_THREADS;
    const int jb = tid * C, je = min(B, jb + C);
    int cp = 0, cn = 0;
    for (int j = jb; j < je; ++j) {
        const int32_t lj = labels[j];
        cp += (lj == la && j != a);
        cn += (lj != la);
    }
    scan[tid + 1] = cp; scan[TRIP_THREADS + 1 + tid + 1] = cn;
    if (tid == 0) { scan[0] = 0; scan[TRIP_THREADS + 1] = 0; }
    __syncthreads();
    if (tid < 2) {   // tiny serial scans (2 x 256 adds)
        int* s = scan + tid * (TRIP_THREADS + 1);
        for (int k = 1; k <= TRIP_THREADS; ++k) s[k] += s[k - 1];
    }
    __syncthreads();
    const int nP = scan[TRIP_THREADS], nN = scan[TRIP_THREADS + 1 + TRIP_THREADS];
    {
        int op = scan[tid], on = scan[TRIP_THREADS + 1 + tid];
        for (int j = jb; j < je; ++j) {
            const int32_t lj = labels[j];
            float d = 0.f;
            for (int s = 0; s < d_splits; ++s) d += D_slabs[(int64_t)s * slab_stride + (int64_t)a * ldd + j];
            if (lj == la) { if (j != a) { pu[op] = d; pidx[op] = j; ++op; } }
            else { nv[on] = d; nidx[on] = j; ++on; }
        }
    }
    __syncthreads();

    // ---- role 1: each thread owns up to 4 negatives per sweep, loops over all positives ----
    float loss = 0.f;
    unsigned cnt = 0u;
    for (int k0 = 0; k0 < nN; k0 += 4 * TRIP_THREADS) {
        float v[4], gs[4];
        unsigned rc[4];
        bool act[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + q * TRIP_THREADS + tid;
            act[q] = k < nN;
            v[q] = act[q] ? nv[k] : 0.f;
            gs[q] = 0.f; rc[q] = 0u;
        }
        for (int p = 0; p < nP; ++p) {
            const float u = pu[p];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = v[q] - u;                    // triplet_distance[a,p,n] (:106)
                float sp, sg;
                softplus_sigmoid(t, sp, sg);
                const bool pos = t > 1e-16f;                 // (:114)
                const bool use = act[q] && (POS_ONLY ? pos : true);
                loss += use ? sp : 0.f;
                gs[q] += use ? sg : 0.f;
                cnt += (act[q] && pos) ? 1u : 0u;
                if (POS_ONLY) rc[q] += (act[q] && pos) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + q * TRIP_THREADS + tid;
            if (act[q]) { Grow[nidx[k]] = gs[q]; if (POS_ONLY) Rrow[nidx[k]] = rc[q]; }
        }
    }
    // ---- role 2: each thread owns up to 4 positives, loops over all negatives (sigmoid only) ----
    for (int k0 = 0; k0 < nP; k0 += 4 * TRIP_THREADS) {
        float u[4], gs[4];
        unsigned rc[4];
        bool act[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + q * TRIP_THREADS + tid;
            act[q] = k < nP;
            u[q] = act[q] ? pu[k] : 0.f;
            gs[q] = 0.f; rc[q] = 0u;
        }
        for (int n = 0; n < nN; ++n) {
            const float vv = nv[n];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = vv - u[q];
                const float sg = sigmoid_only(t);
                const bool pos = t > 1e-16f;
                const bool use = act[q] && (POS_ONLY ? pos : true);
                gs[q] += use ? sg : 0.f;
                if (POS_ONLY) rc[q] += (act[q] && pos) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + q * TRIP_THREADS + tid;
            if (act[q]) { Grow[pidx[k]] = -gs[q]; if (POS_ONLY) Rrow[pidx[k]] = rc[q]; }
        }
    }
    const float ltot = block_sum_f(loss, red);
    const unsigned ctot = block_sum_u(cnt, redu);
    if (tid == 0) { loss_part[a] = ltot; npos_part[a] = ctot; }
}

// ------------------------------------------------------------------------------------------------
// batch_hard
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TRIP_THREADS) void batch_hard_kernel(const float* __restrict__ D_slabs, int d_splits,
                                                                  int64_t slab_stride, int64_t ldd,
                                                                  const int32_t* __restrict__ labels, int B, int Bp,
                                                                  float* __restrict__ loss_part, uint32_t* __restrict__ cnt_part,
                                                                  int32_t* __restrict__ dw, float* __restrict__ G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* drow = reinterpret_cast<float*>(smem);            // [Bp]
    float* red = drow + Bp;                                   // [4]
    unsigned* redu = reinterpret_cast<unsigned*>(red + 4);    // [4]
    const int a = blockIdx.x, tid = threadIdx.x;
    const int32_t la = labels[a];
    float* Grow = G + (int64_t)a * Bp;

    float mx = -INFINITY;
    for (int j = tid; j < B; j += TRIP_THREADS) {
        float d = 0.f;
        for (int s = 0; s < d_splits; ++s) d += D_slabs[(int64_t)s * slab_stride + (int64_t)a * ldd + j];
        drow[j] = d;
        mx = fmaxf(mx, d);
    }
    const float rowmax = block_max_f(mx, red);                               // :227
    float mn = INFINITY, hnl = -INFINITY;
    for (int j = tid; j < B; j += TRIP_THREADS) {
        const int32_t lj = labels[j];
        const float ap = (lj == la && j != a) ? 1.f : 0.f;                   // :223-224
        const float an = (lj != la) ? 1.f : 0.f;                             // :236-237
        const float d = drow[j];
        mn = fminf(mn, d + rowmax * (1.0f - ap));                            // :228
        hnl = fmaxf(hnl, an * d);                                            // :240  (invalid -> 0, not -inf)
    }
    const float hp = block_min_f(mn, red);                                   // :231
    const float hn = block_max_f(hnl, red);                                  // :243
    const float dist = fmaxf(hn - hp, 0.f);                                  // :247
    const bool cnt = dist > 0.f;                                             // :249
    // tie counts for the gradient split (TF reduce_min/max gradient = indicator / num_selected)
    unsigned c_n = 0, c_p = 0, c_m = 0, c_pinv = 0;
    for (int j = tid; j < B; j += TRIP_THREADS) {
        const int32_t lj = labels[j];
        const float ap = (lj == la && j != a) ? 1.f : 0.f;
        const float an = (lj != la) ? 1.f : 0.f;
        const float d = drow[j];
        const bool tp = (d + rowmax * (1.0f - ap)) == hp;
        c_n += (an * d == hn);
        c_p += tp;
        c_pinv += (tp && ap == 0.f);
        c_m += (d == rowmax);
    }
    const unsigned n_n = block_sum_u(c_n, redu), n_p = block_sum_u(c_p, redu);
    const unsigned n_m = block_sum_u(c_m, redu), n_pinv = block_sum_u(c_pinv, redu);
    const float gd = cnt ? sigmoid_only(dist) : 0.f;                         // softplus' = sigmoid; * triplet_count
    const float g_n = gd / (float)n_n;
    const float g_p = -gd / (float)n_p;
    const float g_m = (g_p * (float)n_pinv) / (float)n_m;                    // gradient through the row-max shift
    for (int j = tid; j < Bp; j += TRIP_THREADS) {
        float g = 0.f;
        if (j < B) {
            const int32_t lj = labels[j];
            const float ap = (lj == la && j != a) ? 1.f : 0.f;
            const float an = (lj != la) ? 1.f : 0.f;
            const float d = drow[j];
            if (an * d == hn) g += an * g_n;
            if ((d + rowmax * (1.0f - ap)) == hp) g += g_p;
            if (d == rowmax) g += g_m;
            if (cnt) {                                                       // data_weight :251-253
                const int w = (d == hp) + (d == hn) + (j == a);
                if (w) atomicAdd(&dw[j], w);
            }
        }
        Grow[j] = g;
    }
    if (tid == 0) {
        loss_part[a] = cnt ? softplus_tf(dist) : 0.f;                        // :256
        cnt_part[a] = cnt ? 1u : 0u;
    }
}

}  // namespace dae

using namespace dae;

extern "C" int dae_triplet_batch_all(const float* D_slabs, int32_t d_splits, int64_t slab_stride, int64_t ldd,
                                     const int32_t* labels, int32_t B, int32_t Bp, int32_t pos_only, float* loss_part,
                                     uint32_t* npos_part, float* G, uint32_t* role_cnt, void* stream) {
    DAE_CHECK_ARG(D_slabs && labels && loss_part && npos_part && G, "batch_all: null input");
    DAE_CHECK_ARG(B > 0 && B <= Bp && Bp <= TRIP_MAX_B, "batch_all: batch %d (padded %d) exceeds the supported %d", B, Bp, TRIP_MAX_B);
    DAE_CHECK_ARG(!pos_only || role_cnt, "batch_all: role_cnt required with pos_triplets_only");
    const size_t lds = (size_t)Bp * 16 + 2 * (TRIP_THREADS + 1) * sizeof(int) + 8 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(batch_all_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, TRIP_MAX_B * 16 + 4096));
        DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(batch_all_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, TRIP_MAX_B * 16 + 4096));
        attr_done = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (pos_only)
        hipLaunchKernelGGL((batch_all_kernel<true>), dim3(B), dim3(TRIP_THREADS), lds, st, D_slabs, d_splits, slab_stride, ldd,
                           labels, B, Bp, loss_part, npos_part, G, role_cnt);
    else
        hipLaunchKernelGGL((batch_all_kernel<false>), dim3(B), dim3(TRIP_THREADS), lds, st, D_slabs, d_splits, slab_stride, ldd,
                           labels, B, Bp, loss_part, npos_part, G, role_cnt);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_triplet_batch_hard(const float* D_slabs, int32_t d_splits, int64_t slab_stride, int64_t ldd,
                                      const int32_t* labels, int32_t B, int32_t Bp, float* loss_part, uint32_t* cnt_part,
                                      int32_t* dw, float* G, void* stream) {
    DAE_CHECK_ARG(D_slabs && labels && loss_part && cnt_part && dw && G, "batch_hard: null input");
    DAE_CHECK_ARG(B > 0 && B <= Bp && Bp <= 4 * TRIP_MAX_B, "batch_hard: batch %d too large", B);
    hipStream_t st = (hipStream_t)stream;
    DAE_CHECK_HIP(hipMemsetAsync(dw, 0, (size_t)Bp * sizeof(int32_t), st));
    const size_t lds = (size_t)Bp * 4 + 64;
    hipLaunchKernelGGL(batch_hard_kernel, dim3(B), dim3(TRIP_THREADS), lds, st, D_slabs, d_splits, slab_stride, ldd, labels, B, Bp,
                       loss_part, cnt_part, dw, G);
    DAE_CHECK_LAUNCH();
    return 0;
}
