// dae_elementwise.hip -- the small HBM-bound kernels between the MFMA GEMMs of the DAE step:
// split-K reductions fused with the non-linear epilogues, label statistics, bias gradients,
// optimizer update (with low-precision shadow refresh) and the per-step statistics.
// All of them are stream-ordered, deterministic (fixed reduction order; only integer atomics).
#include "dae_sym.h"
#include "dae_common.h"
#include "dae_kernels.h"
#include "dae_label.h"

namespace dae {

// ------------------------------------------------------------------------------------------------
// K2 encode epilogue: h = act(sum_s slab_s + bh) - act(bh)          (autoencoder.py:389)
// 64x64 tiles; h^T goes through an LDS transpose so both images are written coalesced.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void encode_finish_kernel(const float* __restrict__ slabs, int splits, int64_t slab_stride,
                                                            int64_t ld_slab, const float* __restrict__ bh, int B, int H,
                                                            int enc_act, float* __restrict__ h_f32, T* __restrict__ h_lo,
                                                            int64_t ldh, T* __restrict__ h_t, int64_t ldht,
                                                            bf16_t* __restrict__ hcat_a, bf16_t* __restrict__ hcat_b, int Hp,
                                                            T* __restrict__ h_t2) {
    __shared__ float tile[32][65];
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = j0 + tx;
    const float b = bh[j];
    const float ab = act_apply(enc_act, b);
    float z[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = 0.f;
    // slabs in groups of 4: 32 independent loads in flight per thread (the kernel is latency-bound: 224 blocks, one HBM
    // round trip per group instead of one per slab); the summation order over slabs is unchanged
    for (int s0 = 0; s0 < splits; s0 += 4) {
        float t[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* sl = slabs + (int64_t)(s0 + u) * slab_stride + (int64_t)i0 * ld_slab + j;
#pragma unroll
            for (int k = 0; k < 8; ++k) t[u][k] = (s0 + u < splits) ? sl[(int64_t)(ty + 4 * k) * ld_slab] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) z[k] += t[u][k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = ty + 4 * k, i = i0 + r;
        const float h = (i < B && j < H) ? act_apply(enc_act, z[k] + b) - ab : 0.f;
        if (h_f32) h_f32[(int64_t)i * ldh + j] = h;
        if (h_lo) h_lo[(int64_t)i * ldh + j] = Elem<T>::from(h);
        if (hcat_a) {   // split-bf16 operands of the Gram matrix: h = hi + lo, D ~= hi.hi + hi.lo + lo.hi (|err| ~ 2^-17 |h|^2)
            const bf16_t hi = f2bf(h);
            const bf16_t lo = f2bf(h - bf2f(hi));
            const int64_t o = (int64_t)i * (3 * Hp) + j;
            hcat_a[o] = hi; hcat_a[o + Hp] = hi; hcat_a[o + 2 * Hp] = lo;
            hcat_b[o] = hi; hcat_b[o + Hp] = lo; hcat_b[o + 2 * Hp] = hi;
        }
        tile[r][tx] = h;
    }
    __syncthreads();
    if (h_t) {                                          // 64 feature rows x 32 batch columns
        const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = r0 + 8 * k;
            h_t[(int64_t)(j0 + r) * ldht + i0 + c] = Elem<T>::from(tile[c][r]);
            if (h_t2) h_t2[(int64_t)(j0 + r) * ldht + i0 + c] = elem_residual<T>(tile[c][r]);      // split-bf16 mode: lo image of h^T
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K8 (middle): dh = sum_s slab_s (+ dh_extra); delta1 = dh * act'(z1); delta1^T; column partial sums
// for db_h = sum_i delta1 - act'(bh) * sum_i dh    (the -act(bh) term of autoencoder.py:389)
// ------------------------------------------------------------------------------------------------
// 512 threads = 64 columns x 8 row groups; a thread owns rows ty + 8k (k < 4) of a 32-row block.  The kernel is pure latency
// (224 blocks, < 20 MB): every load of a thread -- h, dh_extra and up to 8 slabs x 4 rows -- is issued before the first use,
// so the block pays one memory round trip instead of one per slab group.
constexpr int DHF_THREADS = 512;
template <typename T>
__global__ __launch_bounds__(DHF_THREADS) void dh_finish_kernel(const float* __restrict__ slabs, int splits, int64_t slab_stride,
                                                                int64_t ld_slab, const float* __restrict__ dh_extra,
                                                                const float* __restrict__ h_f32, int64_t ldh,
                                                                const float* __restrict__ bh, int B, int H, int enc_act,
                                                                T* __restrict__ delta1_t, int64_t ldt, float* __restrict__ colsum_part,
                                                                int Hp, float* __restrict__ delta1_f32, T* __restrict__ delta1_lo,
                                                                T* __restrict__ delta1_t2, float in_scale, float out_scale) {
    __shared__ float tile[32][65];
    __shared__ float cs[2][8][64];
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = j0 + tx;
    float hv[4], ex[4], dhv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = i0 + ty + 8 * k;
        hv[k] = h_f32[(int64_t)i * ldh + j];
        ex[k] = dh_extra ? dh_extra[(int64_t)i * ldh + j] : 0.f;
        dhv[k] = 0.f;
    }
    const float ab = act_apply(enc_act, bh[j]);
    for (int s0 = 0; s0 < splits; s0 += 8) {            // 8 slabs x 4 rows = 32 loads in flight; slab order of the sum unchanged
        float t[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* sl = slabs + (int64_t)min(s0 + u, splits - 1) * slab_stride + (int64_t)i0 * ld_slab + j;
#pragma unroll
            for (int k = 0; k < 4; ++k) t[u][k] = sl[(int64_t)(ty + 8 * k) * ld_slab];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) dhv[k] += (s0 + u < splits) ? t[u][k] : 0.f;
    }
    float s_d1 = 0.f, s_dh = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = ty + 8 * k, i = i0 + r;
        float dh = dhv[k] * in_scale + ex[k];                       // in_scale: 1 / op_scale of the 16-bit delta2 / Gs images (a power of two)
        const bool ok = (i < B && j < H);
        dh = ok ? dh : 0.f;
        const float a1 = hv[k] + ab;                                // act(z1)
        const float d1 = ok ? dh * act_grad(enc_act, a1) : 0.f;
        s_d1 += d1; s_dh += dh;
        tile[r][tx] = d1;
        if (delta1_f32) delta1_f32[(int64_t)i * ldh + j] = d1;
        if (delta1_lo) delta1_lo[(int64_t)i * ldh + j] = Elem<T>::from(sizeof(T) == 2 ? sat16(d1 * out_scale) : d1);
    }
    cs[0][ty][tx] = s_d1; cs[1][ty][tx] = s_dh;
    __syncthreads();
    if (delta1_t) {
        const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;      // 16 feature rows per pass, 32 batch columns each
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + 16 * k;
            const float d1s = sizeof(T) == 2 ? sat16(tile[c][r] * out_scale) : tile[c][r];      // 16-bit images hold out_scale * delta1
            delta1_t[(int64_t)(j0 + r) * ldt + i0 + c] = Elem<T>::from(d1s);
            if (delta1_t2) delta1_t2[(int64_t)(j0 + r) * ldt + i0 + c] = elem_residual<T>(d1s);      // split mode: the lo image
        }
    }
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        const float v = ((cs[which][0][c] + cs[which][1][c]) + (cs[which][2][c] + cs[which][3][c])) +
                        ((cs[which][4][c] + cs[which][5][c]) + (cs[which][6][c] + cs[which][7][c]));
        // layout [2][n_row_blocks][Hp], n_row_blocks = Bp / 32
        colsum_part[((int64_t)which * gridDim.y + blockIdx.y) * Hp + j0 + c] = v;
    }
}

// Gs = scale * (G + G^T) restricted to [0,B)^2, zero elsewhere      (autodiff of D = h h^T)
template <typename T>
__global__ __launch_bounds__(256) void sym_scale_kernel(const float* __restrict__ G, int B, int Bp,
                                                        const float* __restrict__ tri_scalars, T* __restrict__ Gs, float mul) {
    __shared__ float tile[64][65];
    sym_scale_tile<T>(G, B, Bp, tri_scalars, Gs, blockIdx.x, blockIdx.y, tile, mul);
}

// ------------------------------------------------------------------------------------------------
// label statistics (triplet_loss_utils.py:47-76,110-111,129) -- integer exact
// acc: int64[2] = {S = sum_i (n_i - 1), N_valid}, zeroed by the caller before kernel 1
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void label_count_kernel(const int32_t* __restrict__ labels, int B, int32_t* __restrict__ n_same,
                                                          unsigned long long* __restrict__ acc) {
    __shared__ int32_t lab[1024];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t li = (i < B) ? labels[i] : 0;
    int cnt = 0;
    for (int base = 0; base < B; base += 1024) {
        const int n = min(1024, B - base);
        for (int k = threadIdx.x; k < n; k += blockDim.x) lab[k] = labels[base + k];
        __syncthreads();
        for (int k = 0; k < n; ++k) cnt += (lab[k] == li);
        __syncthreads();
    }
    if (i < B) {
        n_same[i] = cnt;
        atomicAdd(&acc[0], (unsigned long long)(cnt - 1));
        atomicAdd(&acc[1], (unsigned long long)(cnt - 1) * (unsigned long long)(B - cnt));
    }
}

__global__ void label_weight_kernel(const int32_t* __restrict__ n_same, int B, int Bp, int triplet,
                                    const unsigned long long* __restrict__ acc, int64_t* __restrict__ nvalid_out,
                                    int64_t* __restrict__ dw_out, float* __restrict__ cw, float alpha,
                                    float* __restrict__ tri_scalars) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp) return;
    if (triplet == DAE_TRIPLET_NONE) {                    // weighted_loss default weight = ones (:266)
        cw[i] = (i < B) ? 1.0f / ((float)B + 1e-16f) : 0.f;
        return;
    }
    const long long S = (long long)acc[0], NV = (long long)acc[1];
    if (i == 0 && nvalid_out) nvalid_out[0] = NV;
    if (i == 0 && tri_scalars && triplet == DAE_TRIPLET_BATCH_ALL) tri_scalars[0] = alpha / ((float)NV + 1e-16f);
    if (i < B) {
        const long long n = n_same[i];
        const long long dw = 2 * (n - 1) * (B - n) + (S - n * (n - 1));
        if (dw_out) dw_out[i] = dw;
        // sum_i dw_i = 3 * N_valid exactly (each valid triplet has three roles)
        if (triplet == DAE_TRIPLET_BATCH_ALL) cw[i] = (float)dw / ((float)(3 * NV) + 1e-16f);
    } else if (triplet == DAE_TRIPLET_BATCH_ALL) {
        cw[i] = 0.f;
    }
}

__global__ __launch_bounds__(1024) void label_stats_small_kernel(LabelJob j) {
    __shared__ __attribute__((aligned(16))) char smem[LABEL_SMEM_BYTES];
    label_stats_block<1024>(j, smem);
}

// ------------------------------------------------------------------------------------------------
// miner partials -> normalisers / statistics (single 1024-thread block)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_d(double v, double* sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if (t < o) sm[t] += sm[t + o];
        __syncthreads();
    }
    double r = sm[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(1024) void triplet_finalize_kernel(int triplet, int pos_only, int B, int Bp, float alpha,
                                                                const float* __restrict__ loss_part,
                                                                const uint32_t* __restrict__ cnt_part,
                                                                const int64_t* __restrict__ nvalid,
                                                                const int32_t* __restrict__ dw_i32,
                                                                const uint32_t* __restrict__ role_cnt,
                                                                float* __restrict__ dw_f32_out, float* __restrict__ cw,
                                                                float* __restrict__ tri_scalars) {
    __shared__ double sm[1024];
    const int t = threadIdx.x;
    double ls = 0.0, cs = 0.0;
    for (int i = t; i < B; i += blockDim.x) { ls += (double)loss_part[i]; cs += (double)cnt_part[i]; }
    const double loss_sum = block_sum_d(ls, sm);
    const double cnt_sum = block_sum_d(cs, sm);
    double N, frac, num = cnt_sum;
    if (triplet == DAE_TRIPLET_BATCH_ALL) {
        const double nv = (double)nvalid[0];
        N = pos_only ? cnt_sum : nv;
        frac = (double)((float)cnt_sum / ((float)nv + 1e-16f));
    } else {
        N = cnt_sum;
        frac = (double)((float)cnt_sum / (float)B);
    }
    const float Nf = (float)N;
    if (t == 0) {
        tri_scalars[0] = alpha / (Nf + 1e-16f);
        tri_scalars[1] = (float)loss_sum / (Nf + 1e-16f);
        tri_scalars[2] = (float)frac;
        tri_scalars[3] = (float)num;
    }
    const bool need_dw = (triplet == DAE_TRIPLET_BATCH_HARD) || pos_only;
    if (!need_dw) return;
    // data_weight from the miner's integer counts, then cw = dw / (sum dw + 1e-16)
    double ws = 0.0;
    for (int j = t; j < B; j += blockDim.x) {
        long long w;
        if (triplet == DAE_TRIPLET_BATCH_HARD) {
            w = dw_i32[j];
        } else {
            w = cnt_part[j];                                   // anchor role
            for (int a = 0; a < B; ++a) w += role_cnt[(int64_t)a * Bp + j];   // positive + negative roles
        }
        dw_f32_out[j] = (float)w;
        ws += (double)w;
    }
    const double wsum = block_sum_d(ws, sm);
    const float wsf = (float)wsum + 1e-16f;
    for (int j = t; j < Bp; j += blockDim.x) cw[j] = (j < B) ? dw_f32_out[j] / wsf : 0.f;
}

// cosine_proximity: reduce the first pass' partials into per-row statistics and the row loss.
// One wave per 64 rows x one eighth of the partial rows at a time would still be a serial chain of n_col_waves (316 at
// F = 10000) dependent-free but strided loads per thread; instead a workgroup owns 64 rows and its 4 waves split the partial
// rows (coalesced 256-byte reads, 8 loads in flight), combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void cos_reduce_kernel(const float* __restrict__ cos_part, int n_col_waves, int B, int Bp,
                                                         float* __restrict__ cos_stats, float* __restrict__ rowloss) {
    __shared__ float sm[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float yy = 0.f, xy = 0.f;
    if (i < Bp) {
        const int per = (n_col_waves + 3) / 4;
        const int p0 = w * per, p1 = min(n_col_waves, p0 + per);
#pragma unroll 8
        for (int p = p0; p < p1; ++p) {
            yy += cos_part[(int64_t)p * Bp + i];
            xy += cos_part[(int64_t)(n_col_waves + p) * Bp + i];
        }
    }
    sm[0][w][lane] = yy; sm[1][w][lane] = xy;
    __syncthreads();
    if (w == 0 && i < Bp) {
        yy = (sm[0][0][lane] + sm[0][1][lane]) + (sm[0][2][lane] + sm[0][3][lane]);
        xy = (sm[1][0][lane] + sm[1][1][lane]) + (sm[1][2][lane] + sm[1][3][lane]);
        cos_stats[Bp + i] = yy;
        cos_stats[2 * Bp + i] = xy;
        rowloss[i] = (i < B) ? -xy * rsqrtf(fmaxf(yy, 1e-12f)) : 0.f;    // -sum xhat*yhat (:273)
    }
}

// bias gradients into the flat gradient buffer
// optimizer element update (autoencoder.py:444-477, tf.train.* semantics)
__device__ __forceinline__ float opt_update(int opt, float lr, float mom, float p, float g, float* s1, float* s2, int64_t k) {
    switch (opt) {
        case DAE_OPT_SGD: return p - lr * g;
        case DAE_OPT_ADAGRAD: { float a = s1[k] + g * g; s1[k] = a; return p - lr * g * rsqrtf(a); }
        case DAE_OPT_MOMENTUM: { float a = mom * s1[k] + g; s1[k] = a; return p - lr * a; }
        default: {   // Adam; lr already holds lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
            float m = 0.9f * s1[k] + 0.1f * g;
            float v = 0.999f * s2[k] + 0.001f * g * g;
            s1[k] = m; s2[k] = v;
            return p - lr * m / (sqrtf(v) + 1e-8f);
        }
    }
}


// apply != 0 also performs the optimizer update of the biases (slot layout [bh (Hp) | bv (Fp)] at s1b / s2b)
__device__ __forceinline__ void bias_grads_body(const BiasArgs& a, int k) {
    if (k < a.Fp) {
        float s = 0.f;
        if (k < a.F) {
#pragma unroll 8
            for (int p = 0; p < a.n_row_waves; ++p) s += a.dbv_part[(int64_t)p * a.Fp + k];
        }
        a.dbv[k] = s;
        if (a.apply) a.bv[k] = opt_update(a.opt, a.lr, a.mom, a.bv[k], s * a.gscale, a.s1b, a.s2b, (int64_t)a.Hp + k);
    } else if (k < a.Fp + a.Hp) {
        const int j = k - a.Fp;
        float s1 = 0.f, s2 = 0.f;
        if (j < a.H) {
#pragma unroll 8
            for (int p = 0; p < a.n_row_blocks; ++p) {
                s1 += a.colsum_part[(int64_t)p * a.Hp + j];
                s2 += a.colsum_part[((int64_t)a.n_row_blocks + p) * a.Hp + j];
            }
            const float ab = act_apply(a.enc_act, a.bh[j]);
            s1 -= act_grad(a.enc_act, ab) * s2;
        }
        a.dbh[j] = s1;
        if (a.apply) a.bh[j] = opt_update(a.opt, a.lr, a.mom, a.bh[j], s1 * a.gscale, a.s1b, a.s2b, (int64_t)j);
    }
}
__global__ void bias_grads_kernel(BiasArgs a) { bias_grads_body(a, blockIdx.x * blockDim.x + threadIdx.x); }

// ------------------------------------------------------------------------------------------------
// K9 optimizer (autoencoder.py:444-477, tf.train.* semantics)
// ------------------------------------------------------------------------------------------------
// 64 x 64 tiles, 256 threads; a thread owns 4 consecutive columns of rows r0, r0 + 16, r0 + 32, r0 + 48: master weights, gradient and optimizer slots move
// as 16-byte pieces of 256-byte row runs, the row-major shadow(s) as 8-byte pieces; the transposed shadow(s) leave through an LDS tile as 8-byte pieces of
// 128-byte runs.  (The element-per-thread form of this kernel moved ~1 GB in 202 us at F = 50000: half the HBM rate of the dW epilogue's block-wise form.)
template <typename T>
__global__ __launch_bounds__(256) void opt_w_kernel(int opt, float lr, float mom, float gscale, float* __restrict__ W,
                                                    const float* __restrict__ grad, float* __restrict__ s1,
                                                    float* __restrict__ s2, int ldwt, int Hp, T* __restrict__ W_lo,
                                                    T* __restrict__ Wt_lo, int apply, T* __restrict__ W_lo2, T* __restrict__ Wt_lo2) {
    // W_lo2 / Wt_lo2 (split mode): the lo images, lo16(W - hi16(W)) in both layouts.  ldwt: leading dimension of the transposed shadows (Fp); the pointers
    // may address a row band of W (launch_opt_step's f0 / f1): blockIdx.y counts 64-row blocks from the band's first row
    __shared__ float tile[64][65];
    const int j0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
    const int c4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i;
        const int64_t k = (int64_t)(f0 + r) * Hp + j0 + c4 * 4;
        f32x4 p = *reinterpret_cast<const f32x4*>(W + k);
        if (apply) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(grad + k) * gscale;
            if (opt == DAE_OPT_SGD) {
                p = p - lr * g;
            } else if (opt == DAE_OPT_ADAGRAD) {
                f32x4 a = *reinterpret_cast<const f32x4*>(s1 + k) + g * g;
                *reinterpret_cast<f32x4*>(s1 + k) = a;
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = p[j] - lr * g[j] * rsqrtf(a[j]);
            } else if (opt == DAE_OPT_MOMENTUM) {
                const f32x4 a = mom * *reinterpret_cast<const f32x4*>(s1 + k) + g;
                *reinterpret_cast<f32x4*>(s1 + k) = a;
                p = p - lr * a;
            } else {             // Adam; lr already holds lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
                const f32x4 m = 0.9f * *reinterpret_cast<const f32x4*>(s1 + k) + 0.1f * g;
                const f32x4 v = 0.999f * *reinterpret_cast<const f32x4*>(s2 + k) + 0.001f * g * g;
                *reinterpret_cast<f32x4*>(s1 + k) = m; *reinterpret_cast<f32x4*>(s2 + k) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = p[j] - lr * m[j] / (sqrtf(v[j]) + 1e-8f);
            }
            *reinterpret_cast<f32x4*>(W + k) = p;
        }
        if constexpr (sizeof(T) == 2) {
            if (W_lo) { uint2 v; v.x = f2bf_pack_hw(p[0], p[1]); v.y = f2bf_pack_hw(p[2], p[3]); *reinterpret_cast<uint2*>(W_lo + k) = v; }
            if (W_lo2) { uint2 v; v.x = bf_residual_pack_hw(p[0], p[1]); v.y = bf_residual_pack_hw(p[2], p[3]); *reinterpret_cast<uint2*>(W_lo2 + k) = v; }
        } else {
            if (W_lo) *reinterpret_cast<f32x4*>(W_lo + k) = p;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[r][c4 * 4 + j] = p[j];
    }
    __syncthreads();
    if (Wt_lo) {                                        // transposed tile: thread (c4, r0) writes features 4 c4 .. + 3 of hidden rows r0 + 16 i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 16 * i;
            const float a = tile[c4 * 4][r], b = tile[c4 * 4 + 1][r], c = tile[c4 * 4 + 2][r], d = tile[c4 * 4 + 3][r];
            const int64_t kt = (int64_t)(j0 + r) * ldwt + f0 + c4 * 4;
            if constexpr (sizeof(T) == 2) {
                uint2 v; v.x = f2bf_pack_hw(a, b); v.y = f2bf_pack_hw(c, d);
                *reinterpret_cast<uint2*>(Wt_lo + kt) = v;
                if (Wt_lo2) { uint2 w; w.x = bf_residual_pack_hw(a, b); w.y = bf_residual_pack_hw(c, d); *reinterpret_cast<uint2*>(Wt_lo2 + kt) = w; }
            } else {
                f32x4 v = {a, b, c, d};
                *reinterpret_cast<f32x4*>(Wt_lo + kt) = v;
            }
        }
    }
}

// Wt_lo = W_lo^T (64 x 64 tiles through LDS): after a sharded optimizer step + all-gather of W_lo every rank rebuilds the
// transposed shadow locally instead of receiving it (halves the all-gather)
template <typename T>
__global__ __launch_bounds__(256) void transpose_lo_kernel(const T* __restrict__ W_lo, int Fp, int Hp, T* __restrict__ Wt_lo) {
    __shared__ T tile[64][66];
    const int j0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) tile[r][tx] = W_lo[(int64_t)(f0 + r) * Hp + j0 + tx];
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) Wt_lo[(int64_t)(j0 + r) * Fp + f0 + tx] = tile[tx][r];
}

// Data-parallel step, after the all-gather: `recv` holds one chunk per rank, each = [chunk_rows x Hp low-precision rows of W owned by
// that rank | ... | that rank's LOCAL bias gradients (Hp + Fp floats) at bias_off].  Blocks (x, y < Fp / 64) copy a 64 x 64 tile of rows
// into the contiguous shadow W_lo AND write its transpose into Wt_lo (the rebuild that used to be its own launch after a separate copy);
// blocks y >= Fp / 64 sum the ranks' bias gradients in rank order (identical on every rank) and apply the optimizer to bh / bv.
template <typename T>
__global__ __launch_bounds__(256) void dp_unpack_kernel(const char* __restrict__ recv, int world, int chunk_rows, int64_t chunk_stride,
                                                        int64_t bias_off, int Fp, int Hp, T* __restrict__ W_lo, T* __restrict__ Wt_lo,
                                                        int opt, float lr, float mom, float gscale, float* __restrict__ bh, float* __restrict__ bv,
                                                        float* __restrict__ s1b, float* __restrict__ s2b, float* __restrict__ grad_b) {
    __shared__ T tile[64][66];
    const int ty_tiles = Fp / 64;
    if ((int)blockIdx.y >= ty_tiles) {
        const int k = (((int)blockIdx.y - ty_tiles) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        if (k >= Hp + Fp) return;
        float g = 0.f;
        for (int r = 0; r < world; ++r) g += reinterpret_cast<const float*>(recv + (int64_t)r * chunk_stride + bias_off)[k];
        grad_b[k] = g;                                        // the rank-summed bias gradient stays readable (Engine.grads)
        float* q = (k < Hp) ? &bh[k] : &bv[k - Hp];
        *q = opt_update(opt, lr, mom, *q, g * gscale, s1b, s2b, k);
        return;
    }
    const int j0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
    const int r = f0 / chunk_rows;                            // chunk_rows % 64 == 0: a tile never straddles two ranks' chunks
    const T* src = reinterpret_cast<const T*>(recv + (int64_t)r * chunk_stride) + (int64_t)(f0 - r * chunk_rows) * Hp;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        const T v = src[(int64_t)i * Hp + j0 + tx];
        tile[i][tx] = v;
        W_lo[(int64_t)(f0 + i) * Hp + j0 + tx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) Wt_lo[(int64_t)(j0 + i) * Fp + f0 + tx] = tile[tx][i];
}

__global__ void opt_bias_kernel(int opt, float lr, float mom, float gscale, float* __restrict__ bh, float* __restrict__ bv,
                                const float* __restrict__ grad_b, float* __restrict__ s1b, float* __restrict__ s2b, int Hp,
                                int Fp) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Hp + Fp) return;
    float* p = (k < Hp) ? &bh[k] : &bv[k - Hp];
    *p = opt_update(opt, lr, mom, *p, grad_b[k] * gscale, s1b, s2b, k);
}

// per-step statistics (autoencoder.py:233 fetch list); any block size (sm holds blockDim.x doubles)
// three block sums at once: xor-butterfly inside each wave (fixed order), one LDS round over the waves (summed in wave order)
__device__ __forceinline__ void block_sum_d3(double& x, double& y, double& z, double* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); z += __shfl_xor(z, o); }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { sm[3 * w] = x; sm[3 * w + 1] = y; sm[3 * w + 2] = z; }
    __syncthreads();
    x = y = z = 0.0;
    for (int i = 0; i < nw; ++i) { x += sm[3 * i]; y += sm[3 * i + 1]; z += sm[3 * i + 2]; }
    __syncthreads();
}

__device__ __forceinline__ void step_stats_body(const StatsArgs& a, double* sm) {
    // batch_all (all valid triplets): fold triplet_finalize in -- loss = sum/(N_valid+1e-16), num = sum of counts.
    // One workgroup on the step's critical path: every load is issued before the (single) reduction round.
    double lsum = 0.0, csum = 0.0, s = 0.0;
    const float nvf = (a.loss_part && a.nvalid) ? (float)a.nvalid[0] : 0.f;
    if (a.loss_part)
        for (int i = threadIdx.x; i < a.B; i += blockDim.x) { lsum += (double)a.loss_part[i]; csum += (double)a.cnt_part[i]; }
    if (a.tile_part) {                                    // per-tile weighted sums from the fused decode epilogue
        for (int i = threadIdx.x; i < a.n_tiles; i += blockDim.x) s += (double)a.tile_part[i];
    } else {
        for (int i = threadIdx.x; i < a.B; i += blockDim.x) {
            float r = 0.f;
            for (int p = 0; p < a.n_col_waves; ++p) r += a.rowloss_part[(int64_t)p * a.Bp + i];
            s += (double)(r * a.cw[i]);
        }
    }
    block_sum_d3(lsum, csum, s, sm);
    if (threadIdx.x == 0) {
        const float aef = (float)s;
        float tl = 0.f, fr = 0.f, nm = 0.f, nv = 0.f;
        if (a.triplet != DAE_TRIPLET_NONE) {
            if (a.loss_part) {
                tl = (float)lsum / (nvf + 1e-16f); nm = (float)csum; fr = nm / (nvf + 1e-16f);
                if (a.tri_scalars) { a.tri_scalars[1] = tl; a.tri_scalars[2] = fr; a.tri_scalars[3] = nm; }
            } else {
                tl = a.tri_scalars[1]; fr = a.tri_scalars[2]; nm = a.tri_scalars[3];
            }
            nv = (a.triplet == DAE_TRIPLET_BATCH_ALL && a.nvalid) ? (float)a.nvalid[0] : nm;
        }
        a.stats[DAE_STAT_COST] = (a.triplet != DAE_TRIPLET_NONE) ? aef + a.alpha * tl : aef;
        a.stats[DAE_STAT_AE] = aef;
        a.stats[DAE_STAT_TRIPLET] = tl;
        a.stats[DAE_STAT_FRACTION] = fr;
        a.stats[DAE_STAT_NUM] = nm;
        a.stats[DAE_STAT_NVALID] = nv;
        a.stats[6] = 0.f; a.stats[7] = 0.f;
    }
}
__global__ __launch_bounds__(1024) void step_stats_kernel(StatsArgs a) {
    __shared__ double sm[1024];
    step_stats_body(a, sm);
}

// Tail of a training step in ONE launch (each of these was a separate ~5 us launch):
//   blocks [0, nb_bias)        bias-gradient reduction (+ bias optimizer update)
//   block  nb_bias             the step's statistics
//   blocks (nb_bias, ...)      un-scatter: zero the entries of x~^T this step's gather wrote (one wave per batch row),
//                              so the next step needs no 18 MB memset of the transposed operand
__global__ __launch_bounds__(256) void step_tail_kernel(BiasArgs ba, StatsArgs sa, ClearArgs ca, int nb_bias, int stats_on) {
    __shared__ double sm[256];
    const int b = blockIdx.x;
    if (b < nb_bias) { bias_grads_body(ba, b * 256 + threadIdx.x); return; }
    if (b == nb_bias) { if (stats_on) step_stats_body(sa, sm); return; }
    const int i = (b - nb_bias - 1) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= ca.B) return;
    const int64_t row = ca.row_idx[i];
    const int64_t s0 = ca.indptr[row], e0 = ca.indptr[row + 1];
    // 256 entries per round: the four index loads of a lane are issued together, then the four stores (one load latency per round
    // instead of four -- the stores may alias the index array as far as the compiler knows, so it would not hoist the loads itself)
    for (int64_t k0 = s0; k0 < e0; k0 += 256) {
        int col[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t k = k0 + lane + 64 * j;
            col[j] = k < e0 ? ca.indices[k] : ca.F;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (col[j] < ca.F) {
                if (ca.xtb) ca.xtb[(int64_t)col[j] * ca.ldxt + (i >> 5)] = 0u;        // every lane that touches the word writes the same zero
                else if (ca.es == 2) {
                    const int64_t o = ca.rm ? (int64_t)i * ca.ldt + col[j] : (int64_t)col[j] * ca.ldt + i;
                    reinterpret_cast<bf16_t*>(ca.xct)[o] = 0;
                    if (ca.xct2) reinterpret_cast<bf16_t*>(ca.xct2)[o] = 0;
                }
                else reinterpret_cast<float*>(ca.xct)[ca.rm ? (int64_t)i * ca.ldt + col[j] : (int64_t)col[j] * ca.ldt + i] = 0.f;
            }
        }
    }
}

// explicit (anchor,pos,neg) triplet term (autoencoder_triplet.py:308-311); one wave per triplet row
__global__ __launch_bounds__(256) void explicit_triplet_kernel(const float* __restrict__ h3, int64_t ldh, int B, int H,
                                                               float alpha, float* __restrict__ dh3,
                                                               float* __restrict__ loss_part) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= B) return;
    const float* ho = h3 + (int64_t)i * ldh;
    const float* hp = h3 + (int64_t)(B + i) * ldh;
    const float* hn = h3 + (int64_t)(2 * B + i) * ldh;
    float t = 0.f;
    for (int j = lane; j < H; j += 64) t += ho[j] * hn[j] - ho[j] * hp[j];
    t = wave_sum(t);
    const float g = alpha * sigmoidf_(t) / (float)B;            // d/dt of alpha * mean softplus(t)
    if (lane == 0) loss_part[i] = softplus_tf(t);
    for (int j = lane; j < H; j += 64) {
        const float o = ho[j], p = hp[j], n = hn[j];
        dh3[(int64_t)i * ldh + j] = g * (n - p);
        dh3[(int64_t)(B + i) * ldh + j] = -g * o;
        dh3[(int64_t)(2 * B + i) * ldh + j] = g * o;
    }
}

__global__ __launch_bounds__(1024) void explicit_triplet_finalize_kernel(const float* __restrict__ loss_part, int B,
                                                                         float* __restrict__ tri_scalars) {
    __shared__ double sm[1024];
    double s = 0.0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) s += (double)loss_part[i];
    const double tot = block_sum_d(s, sm);
    if (threadIdx.x == 0) {
        tri_scalars[0] = 0.f;
        tri_scalars[1] = (float)(tot / (double)B);
        tri_scalars[2] = 0.f; tri_scalars[3] = 0.f;
    }
}

}  // namespace dae

using namespace dae;

#define ST(s) ((hipStream_t)(s))

int dae::launch_encode_finish(const float* slabs, int32_t splits, int64_t slab_stride, int64_t ld_slab, const float* bh,
                              int32_t B, int32_t H, int32_t enc_act, int32_t dtype, float* h_f32, void* h_lo, int64_t ldh,
                              void* h_t, int64_t ldht, void* hcat_a, void* hcat_b, void* h_t2, void* stream) {
    DAE_CHECK_ARG((hcat_a == nullptr) == (hcat_b == nullptr), "encode_finish: hcat_a/hcat_b must be given together");
    DAE_CHECK_ARG(slabs && bh && splits >= 1, "encode_finish: null input");
    DAE_CHECK_ARG(B > 0 && H > 0 && ldh >= dae_pad(H) && (!h_t || ldht >= dae_pad(B)), "encode_finish: bad shape");
    DAE_CHECK_ARG(!h_t2 || (h_t && dtype == DAE_BF16), "encode_finish: the lo image of h^T needs h^T and bf16");
    const int Bp = (int)dae_pad(B), Hp = (int)dae_pad(H);
    dim3 grid(Hp / 64, Bp / 32), block(256);
    if (dtype == DAE_BF16)
        DAE_LAUNCH((encode_finish_kernel<bf16_t>), grid, block, 0, ST(stream), slabs, splits, slab_stride, ld_slab, bh, B,
                           H, enc_act, h_f32, (bf16_t*)h_lo, ldh, (bf16_t*)h_t, ldht, (bf16_t*)hcat_a, (bf16_t*)hcat_b, Hp, (bf16_t*)h_t2);
    else
        DAE_LAUNCH((encode_finish_kernel<float>), grid, block, 0, ST(stream), slabs, splits, slab_stride, ld_slab, bh, B,
                           H, enc_act, h_f32, (float*)h_lo, ldh, (float*)h_t, ldht, (bf16_t*)hcat_a, (bf16_t*)hcat_b, Hp, (float*)nullptr);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_encode_finish(const float* slabs, int32_t splits, int64_t slab_stride, int64_t ld_slab, const float* bh,
                                 int32_t B, int32_t H, int32_t enc_act, int32_t dtype, float* h_f32, void* h_lo, int64_t ldh,
                                 void* h_t, int64_t ldht, void* hcat_a, void* hcat_b, void* stream) {
    return launch_encode_finish(slabs, splits, slab_stride, ld_slab, bh, B, H, enc_act, dtype, h_f32, h_lo, ldh, h_t, ldht, hcat_a, hcat_b, nullptr, stream);
}

int dae::launch_dh_finish(const float* slabs, int splits, int64_t slab_stride, int64_t ld_slab, const float* dh_extra, const float* h_f32,
                          int64_t ldh, const float* bh, int B, int H, int enc_act, int dtype, void* delta1_t, int64_t ldt, float* colsum_part,
                          float* delta1_f32, void* delta1_lo, hipStream_t st, void* delta1_t2, float in_scale, float out_scale) {
    DAE_CHECK_ARG(slabs && h_f32 && bh && colsum_part, "dh_finish: null input");
    DAE_CHECK_ARG(B > 0 && H > 0 && ldh >= dae_pad(H) && splits >= 1, "dh_finish: bad shape");
    const int Bp = (int)dae_pad(B), Hp = (int)dae_pad(H);
    dim3 grid(Hp / 64, Bp / 32), block(DHF_THREADS);
    if (dtype == DAE_BF16)
        DAE_LAUNCH((dh_finish_kernel<bf16_t>), grid, block, 0, st, slabs, splits, slab_stride, ld_slab, dh_extra,
                           h_f32, ldh, bh, B, H, enc_act, (bf16_t*)delta1_t, ldt, colsum_part, Hp, delta1_f32, (bf16_t*)delta1_lo, (bf16_t*)delta1_t2,
                           in_scale, out_scale);
    else
        DAE_LAUNCH((dh_finish_kernel<float>), grid, block, 0, st, slabs, splits, slab_stride, ld_slab, dh_extra,
                           h_f32, ldh, bh, B, H, enc_act, (float*)delta1_t, ldt, colsum_part, Hp, delta1_f32, (float*)delta1_lo, (float*)nullptr, in_scale, 1.f);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_dh_finish(const float* slabs, int32_t splits, int64_t slab_stride, int64_t ld_slab, const float* dh_extra,
                             const float* h_f32, int64_t ldh, const float* bh, int32_t B, int32_t H, int32_t enc_act,
                             int32_t dtype, void* delta1_t, int64_t ldt, float* colsum_part, float* delta1_f32, void* stream) {
    return launch_dh_finish(slabs, splits, slab_stride, ld_slab, dh_extra, h_f32, ldh, bh, B, H, enc_act, dtype, delta1_t, ldt, colsum_part,
                            delta1_f32, nullptr, ST(stream));
}

int dae::launch_sym_scale(const float* G, int B, int Bp, const float* tri_scalars, int dtype, void* Gs, float mul, hipStream_t st) {
    DAE_CHECK_ARG(G && tri_scalars && Gs && Bp % DAE_PAD == 0 && B <= Bp, "sym_scale: bad args");
    dim3 grid(Bp / 64, Bp / 64), block(256);
    if (dtype == DAE_BF16)
        DAE_LAUNCH((sym_scale_kernel<bf16_t>), grid, block, 0, st, G, B, Bp, tri_scalars, (bf16_t*)Gs, mul);
    else
        DAE_LAUNCH((sym_scale_kernel<float>), grid, block, 0, st, G, B, Bp, tri_scalars, (float*)Gs, mul);
    DAE_CHECK_LAUNCH();
    return 0;
}
extern "C" int dae_sym_scale(const float* G, int32_t B, int32_t Bp, const float* tri_scalars, int32_t dtype, void* Gs,
                             void* stream) {
    return launch_sym_scale(G, B, Bp, tri_scalars, dtype, Gs, 1.f, ST(stream));
}

extern "C" int dae_label_stats(const int32_t* labels, int32_t B, int32_t Bp, int32_t triplet, int32_t* n_same_scratch,
                               uint64_t* acc_scratch, int64_t* nvalid_out, int64_t* dw_out, float* cw, float alpha,
                               float* tri_scalars, void* stream) {
    DAE_CHECK_ARG(cw && B > 0 && Bp >= B, "label_stats: bad args");
    DAE_CHECK_ARG(triplet == DAE_TRIPLET_NONE || labels, "label_stats: labels required");
    if (Bp <= 1024) {
        LabelJob j{labels, B, Bp, triplet, nvalid_out, dw_out, cw, alpha, tri_scalars};
        DAE_LAUNCH(label_stats_small_kernel, dim3(1), dim3(1024), 0, ST(stream), j);
        DAE_CHECK_LAUNCH();
        return 0;
    }
    if (triplet != DAE_TRIPLET_NONE) {
        DAE_CHECK_ARG(labels && n_same_scratch && acc_scratch, "label_stats: labels/scratch required");
        DAE_CHECK_HIP(hipMemsetAsync(acc_scratch, 0, 2 * sizeof(uint64_t), ST(stream)));
        DAE_LAUNCH(label_count_kernel, dim3((B + 255) / 256), dim3(256), 0, ST(stream), labels, B, n_same_scratch,
                           (unsigned long long*)acc_scratch);
        DAE_CHECK_LAUNCH();
    }
    DAE_LAUNCH(label_weight_kernel, dim3((Bp + 255) / 256), dim3(256), 0, ST(stream), n_same_scratch, B, Bp, triplet,
                       (const unsigned long long*)acc_scratch, nvalid_out, dw_out, cw, alpha, tri_scalars);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_triplet_finalize(int32_t triplet, int32_t pos_only, int32_t B, int32_t Bp, float alpha,
                                    const float* loss_part, const uint32_t* cnt_part, const int64_t* nvalid,
                                    const int32_t* dw_i32, const uint32_t* role_cnt, float* dw_f32_out, float* cw,
                                    float* tri_scalars, void* stream) {
    DAE_CHECK_ARG(loss_part && cnt_part && tri_scalars, "triplet_finalize: null input");
    DAE_CHECK_ARG(triplet == DAE_TRIPLET_BATCH_ALL || triplet == DAE_TRIPLET_BATCH_HARD, "triplet_finalize: bad strategy");
    DAE_CHECK_ARG(triplet != DAE_TRIPLET_BATCH_ALL || nvalid, "triplet_finalize: nvalid required");
    if (triplet == DAE_TRIPLET_BATCH_HARD) DAE_CHECK_ARG(dw_i32 && dw_f32_out && cw, "triplet_finalize: dw buffers required");
    if (triplet == DAE_TRIPLET_BATCH_ALL && pos_only) DAE_CHECK_ARG(role_cnt && dw_f32_out && cw, "triplet_finalize: role_cnt required");
    DAE_LAUNCH(triplet_finalize_kernel, dim3(1), dim3(1024), 0, ST(stream), triplet, pos_only, B, Bp, alpha, loss_part,
                       cnt_part, nvalid, dw_i32, role_cnt, dw_f32_out, cw, tri_scalars);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_cos_reduce(const float* cos_part, int32_t n_col_waves, int32_t B, int32_t Bp, float* cos_stats,
                              float* rowloss, void* stream) {
    DAE_CHECK_ARG(cos_part && cos_stats && rowloss, "cos_reduce: null input");
    DAE_LAUNCH(cos_reduce_kernel, dim3((Bp + 63) / 64), dim3(256), 0, ST(stream), cos_part, n_col_waves, B, Bp,
                       cos_stats, rowloss);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_bias_grads(const float* dbv_part, int32_t n_row_waves, const float* colsum_part, int32_t n_row_blocks,
                              float* bh, int32_t H, int32_t Hp, int32_t F, int32_t Fp, int32_t enc_act, float* dbh,
                              float* dbv, int32_t apply, int32_t opt, float lr, float momentum, float grad_scale, float* bv,
                              float* s1b, float* s2b, void* stream) {
    DAE_CHECK_ARG(dbv_part && colsum_part && bh && dbh && dbv, "bias_grads: null input");
    if (apply) {
        DAE_CHECK_ARG(bv && opt >= DAE_OPT_SGD && opt <= DAE_OPT_ADAM, "bias_grads: bad optimizer arguments");
        DAE_CHECK_ARG(opt == DAE_OPT_SGD || s1b, "bias_grads: optimizer slot s1 required");
        DAE_CHECK_ARG(opt != DAE_OPT_ADAM || s2b, "bias_grads: optimizer slot s2 required");
    }
    const int n = Fp + Hp;
    BiasArgs ba{dbv_part, n_row_waves, colsum_part, n_row_blocks, bh, H, Hp, F, Fp, enc_act, dbh, dbv, apply, opt, lr, momentum, grad_scale, bv, s1b, s2b};
    DAE_LAUNCH(bias_grads_kernel, dim3((n + 255) / 256), dim3(256), 0, ST(stream), ba);
    DAE_CHECK_LAUNCH();
    return 0;
}

// flat layout of grad / s1 / s2: [W (Fp*Hp) | bh (Hp) | bv (Fp)]
int dae::launch_opt_step(int opt, float lr, float momentum, float grad_scale, float* W, float* bh, float* bv, const float* grad, float* s1,
                         float* s2, int Fp, int Hp, int dtype, void* W_lo, void* Wt_lo, void* W_lo2, void* Wt_lo2, int apply, void* stream, int f0, int f1) {
    // [f0, f1): the row band of W to update / refresh (multiples of 64; f1 < 0: all Fp rows) -- the bucketed data-parallel exchange applies the optimizer
    // band by band while the next band's all-reduce is on the wire (dp.AllReduceExchange); the biases are updated with the LAST band (f1 == Fp)
    const bool skip_bias = (apply == 2);     // apply: 0 refresh shadows only, 1 update W and biases, 2 update W only
    if (apply == 2) apply = 1;
    if (f1 < 0) { f0 = 0; f1 = Fp; }
    DAE_CHECK_ARG(W && Fp % DAE_PAD == 0 && Hp % DAE_PAD == 0, "opt_step: bad args");
    DAE_CHECK_ARG(f0 >= 0 && f0 < f1 && f1 <= Fp && f0 % 64 == 0 && f1 % 64 == 0, "opt_step: row band [%d, %d) outside [0, %d] or not multiples of 64", f0, f1, Fp);
    DAE_CHECK_ARG(opt >= DAE_OPT_SGD && opt <= DAE_OPT_ADAM, "opt_step: unknown optimizer %d", opt);
    if (apply) {
        DAE_CHECK_ARG(grad && bh && bv, "opt_step: null grad/bias");
        DAE_CHECK_ARG(opt == DAE_OPT_SGD || s1, "opt_step: optimizer slot s1 required");
        DAE_CHECK_ARG(opt != DAE_OPT_ADAM || s2, "opt_step: optimizer slot s2 required");
    }
    dim3 grid(Hp / 64, (f1 - f0) / 64), block(256);
    const int64_t ro = (int64_t)f0 * Hp;                  // first element of the band in the row-major images
    const int es = dtype == DAE_BF16 ? 2 : 4;
    auto rows = [&](void* q, int64_t off_elems) -> void* { return q ? (char*)q + off_elems * es : nullptr; };
    if (dtype == DAE_BF16)
        DAE_LAUNCH((opt_w_kernel<bf16_t>), grid, block, 0, ST(stream), opt, lr, momentum, grad_scale, W + ro, grad ? grad + ro : nullptr,
                           s1 ? s1 + ro : nullptr, s2 ? s2 + ro : nullptr, Fp, Hp, (bf16_t*)rows(W_lo, ro), (bf16_t*)rows(Wt_lo, f0), apply,
                           (bf16_t*)rows(W_lo2, ro), (bf16_t*)rows(Wt_lo2, f0));
    else
        DAE_LAUNCH((opt_w_kernel<float>), grid, block, 0, ST(stream), opt, lr, momentum, grad_scale, W + ro, grad ? grad + ro : nullptr,
                           s1 ? s1 + ro : nullptr, s2 ? s2 + ro : nullptr, Fp, Hp, (float*)rows(W_lo, ro), (float*)rows(Wt_lo, f0), apply,
                           (float*)nullptr, (float*)nullptr);
    DAE_CHECK_LAUNCH();
    if (apply && !skip_bias && f1 == Fp) {
        const int64_t off = (int64_t)Fp * Hp;
        DAE_LAUNCH(opt_bias_kernel, dim3((Hp + Fp + 255) / 256), dim3(256), 0, ST(stream), opt, lr, momentum, grad_scale, bh,
                           bv, grad + off, s1 ? s1 + off : nullptr, s2 ? s2 + off : nullptr, Hp, Fp);
        DAE_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int dae_opt_step(int32_t opt, float lr, float momentum, float grad_scale, float* W, float* bh, float* bv,
                            const float* grad, float* s1, float* s2, int32_t Fp, int32_t Hp, int32_t dtype, void* W_lo,
                            void* Wt_lo, int32_t apply, void* stream) {
    return launch_opt_step(opt, lr, momentum, grad_scale, W, bh, bv, grad, s1, s2, Fp, Hp, dtype, W_lo, Wt_lo, nullptr, nullptr, apply, stream);
}

extern "C" int dae_opt_bias(int32_t opt, float lr, float momentum, float grad_scale, float* bh, float* bv, const float* grad_b, float* s1b,
                            float* s2b, int32_t Hp, int32_t Fp, void* stream) {
    DAE_CHECK_ARG(bh && bv && grad_b && opt >= DAE_OPT_SGD && opt <= DAE_OPT_ADAM && (opt == DAE_OPT_SGD || s1b) && (opt != DAE_OPT_ADAM || s2b),
                  "opt_bias: bad args");
    DAE_LAUNCH(opt_bias_kernel, dim3((Hp + Fp + 255) / 256), dim3(256), 0, ST(stream), opt, lr, momentum, grad_scale, bh, bv, grad_b, s1b, s2b, Hp, Fp);
    DAE_CHECK_LAUNCH();
    return 0;
}

// data-parallel second half on a ROW RANGE of W (sharded optimizer): rows [f0, f1) of W / slots / W_lo are updated from
// grad_rows (fp32 [f1-f0 x Hp], already summed over the ranks); Wt_lo is NOT touched (dae_transpose_shadow after the all-gather)
extern "C" int dae_opt_step_rows(int32_t opt, float lr, float momentum, float grad_scale, float* W, const float* grad_rows, float* s1,
                                 float* s2, int32_t Hp, int32_t f0, int32_t f1, int32_t dtype, void* W_lo, void* stream) {
    DAE_CHECK_ARG(W && grad_rows && W_lo && Hp % 64 == 0 && f0 >= 0 && f1 >= f0 && f0 % 64 == 0 && f1 % 64 == 0, "opt_step_rows: bad args");
    DAE_CHECK_ARG(opt >= DAE_OPT_SGD && opt <= DAE_OPT_ADAM && (opt == DAE_OPT_SGD || s1) && (opt != DAE_OPT_ADAM || s2), "opt_step_rows: optimizer slots");
    if (f1 == f0) return 0;
    const int64_t off = (int64_t)f0 * Hp;
    const size_t es = dtype == DAE_BF16 ? 2 : 4;
    dim3 grid(Hp / 64, (f1 - f0) / 64), block(256);
    if (dtype == DAE_BF16)
        DAE_LAUNCH((opt_w_kernel<bf16_t>), grid, block, 0, ST(stream), opt, lr, momentum, grad_scale, W + off, grad_rows, s1 ? s1 + off : nullptr,
                           s2 ? s2 + off : nullptr, 0, Hp, (bf16_t*)((char*)W_lo + off * es), (bf16_t*)nullptr, 1, (bf16_t*)nullptr, (bf16_t*)nullptr);
    else
        DAE_LAUNCH((opt_w_kernel<float>), grid, block, 0, ST(stream), opt, lr, momentum, grad_scale, W + off, grad_rows, s1 ? s1 + off : nullptr,
                           s2 ? s2 + off : nullptr, 0, Hp, (float*)((char*)W_lo + off * es), (float*)nullptr, 1, (float*)nullptr, (float*)nullptr);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_dp_unpack(const void* recv, int32_t world, int32_t chunk_rows, int64_t chunk_stride_bytes, int64_t bias_off_bytes,
                             int32_t Fp, int32_t Hp, int32_t dtype, void* W_lo, void* Wt_lo, int32_t opt, float lr, float momentum,
                             float grad_scale, float* bh, float* bv, float* s1b, float* s2b, float* grad_b, void* stream) {
    DAE_CHECK_ARG(recv && W_lo && Wt_lo && bh && bv && grad_b && world >= 1 && Fp % 64 == 0 && Hp % 64 == 0 && chunk_rows % 64 == 0 && chunk_rows > 0,
                  "dp_unpack: bad args");
    DAE_CHECK_ARG((int64_t)world * chunk_rows >= Fp && bias_off_bytes % 4 == 0 && chunk_stride_bytes % 16 == 0, "dp_unpack: chunks do not cover W");
    DAE_CHECK_ARG(opt >= DAE_OPT_SGD && opt <= DAE_OPT_ADAM && (opt == DAE_OPT_SGD || s1b) && (opt != DAE_OPT_ADAM || s2b), "dp_unpack: optimizer slots");
    const int gx = Hp / 64, nb = (Hp + Fp + 255) / 256;
    dim3 grid(gx, Fp / 64 + (nb + gx - 1) / gx), block(256);
    if (dtype == DAE_BF16)
        DAE_LAUNCH((dp_unpack_kernel<bf16_t>), grid, block, 0, ST(stream), (const char*)recv, world, chunk_rows, chunk_stride_bytes, bias_off_bytes,
                           Fp, Hp, (bf16_t*)W_lo, (bf16_t*)Wt_lo, opt, lr, momentum, grad_scale, bh, bv, s1b, s2b, grad_b);
    else
        DAE_LAUNCH((dp_unpack_kernel<float>), grid, block, 0, ST(stream), (const char*)recv, world, chunk_rows, chunk_stride_bytes, bias_off_bytes,
                           Fp, Hp, (float*)W_lo, (float*)Wt_lo, opt, lr, momentum, grad_scale, bh, bv, s1b, s2b, grad_b);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_transpose_shadow(const void* W_lo, int32_t Fp, int32_t Hp, int32_t dtype, void* Wt_lo, void* stream) {
    DAE_CHECK_ARG(W_lo && Wt_lo && Fp % 64 == 0 && Hp % 64 == 0, "transpose_shadow: bad args");
    dim3 grid(Hp / 64, Fp / 64), block(256);
    if (dtype == DAE_BF16) DAE_LAUNCH((transpose_lo_kernel<bf16_t>), grid, block, 0, ST(stream), (const bf16_t*)W_lo, Fp, Hp, (bf16_t*)Wt_lo);
    else DAE_LAUNCH((transpose_lo_kernel<float>), grid, block, 0, ST(stream), (const float*)W_lo, Fp, Hp, (float*)Wt_lo);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_step_stats(const float* rowloss_part, int32_t n_col_waves, const float* tile_part, int32_t n_tiles,
                              const float* cw, int32_t B, int32_t Bp, int32_t triplet, float alpha, float* tri_scalars,
                              const int64_t* nvalid, const float* loss_part, const uint32_t* cnt_part, float* stats,
                              void* stream) {
    DAE_CHECK_ARG(((rowloss_part && cw) || tile_part) && stats, "step_stats: null input");
    DAE_CHECK_ARG(triplet == DAE_TRIPLET_NONE || tri_scalars || loss_part, "step_stats: tri_scalars or miner partials required");
    DAE_CHECK_ARG(!loss_part || (cnt_part && nvalid && triplet == DAE_TRIPLET_BATCH_ALL), "step_stats: miner partials need cnt_part + nvalid");
    StatsArgs sa{rowloss_part, n_col_waves, tile_part, n_tiles, cw, B, Bp, triplet, alpha, tri_scalars, nvalid, stats, loss_part, cnt_part};
    DAE_LAUNCH(step_stats_kernel, dim3(1), dim3(1024), 0, ST(stream), sa);
    DAE_CHECK_LAUNCH();
    return 0;
}

namespace dae {
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n4) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[k];
        uint2 o;
        o.x = f2bf_pack_hw(v.x, v.y); o.y = f2bf_pack_hw(v.z, v.w);
        reinterpret_cast<uint2*>(dst)[k] = o;
    }
}
}  // namespace dae
int dae::launch_cast_bf16(const float* src, void* dst_bf16, int64_t n, hipStream_t st) {
    DAE_CHECK_ARG(src && dst_bf16 && n % 4 == 0, "cast_bf16: bad args");
    DAE_LAUNCH(cast_bf16_kernel, dim3(2048), dim3(256), 0, st, src, (bf16_t*)dst_bf16, n / 4);
    DAE_CHECK_LAUNCH();
    return 0;
}

int dae::launch_step_tail(const BiasArgs& ba, const StatsArgs* sa, const ClearArgs* ca, hipStream_t st) {
    DAE_CHECK_ARG(ba.dbv_part && ba.colsum_part && ba.bh && ba.dbh && ba.dbv, "step_tail: null bias-gradient input");
    const int nb_bias = (ba.Fp + ba.Hp + 255) / 256;
    const int nclear = ca ? (ca->B + 3) / 4 : 0;
    StatsArgs s0; memset(&s0, 0, sizeof(s0));
    ClearArgs c0; memset(&c0, 0, sizeof(c0));
    DAE_LAUNCH(step_tail_kernel, dim3(nb_bias + 1 + nclear), dim3(256), 0, st, ba, sa ? *sa : s0, ca ? *ca : c0, nb_bias, sa ? 1 : 0);
    DAE_CHECK_LAUNCH();
    return 0;
}

extern "C" int dae_explicit_triplet(const float* h3, int64_t ldh, int32_t B, int32_t H, float alpha, float* dh3,
                                    float* loss_part, float* tri_scalars, void* stream) {
    DAE_CHECK_ARG(h3 && dh3 && loss_part && tri_scalars && B > 0, "explicit_triplet: bad args");
    DAE_LAUNCH(explicit_triplet_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(stream), h3, ldh, B, H, alpha, dh3, loss_part);
    DAE_CHECK_LAUNCH();
    DAE_LAUNCH(explicit_triplet_finalize_kernel, dim3(1), dim3(1024), 0, ST(stream), loss_part, B, tri_scalars);
    DAE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stand-alone weighted_loss(input, decode) for the function-level API (triplet_loss_utils.py:262-277):
// one workgroup per row, fp32 x / y, row loss into rowloss[B]; the weighted mean is taken by step_stats.
// ------------------------------------------------------------------------------------------------
namespace dae {
__global__ __launch_bounds__(256) void weighted_loss_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 const float* __restrict__ y, int64_t ldy, int F,
                                                                 int loss_func, float* __restrict__ rowloss) {
    __shared__ float red[3][4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (int64_t)i * ldx;
    const float* yr = y + (int64_t)i * ldy;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int f = tid; f < F; f += 256) {
        const float xv = xr[f], yv = yr[f];
        if (loss_func == DAE_LOSS_CROSS_ENTROPY) {
            a += -(xv * logf(yv + 1e-16f) + (1.0f - xv) * logf((1.0f - yv) + 1e-16f));
        } else if (loss_func == DAE_LOSS_MEAN_SQUARED) {
            const float d = xv - yv;
            a += d * d;
        } else {
            a += xv * yv; b += xv * xv; c += yv * yv;
        }
    }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = b; red[2][tid >> 6] = c; }
    __syncthreads();
    if (tid == 0) {
        a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        if (loss_func == DAE_LOSS_COSINE) a = -a * rsqrtf(fmaxf(b, 1e-12f)) * rsqrtf(fmaxf(c, 1e-12f));
        rowloss[i] = a;
    }
}
}  // namespace dae

extern "C" int dae_weighted_loss_rows(const float* x, int64_t ldx, const float* y, int64_t ldy, int32_t B, int32_t F,
                                      int32_t loss_func, float* rowloss, void* stream) {
    DAE_CHECK_ARG(x && y && rowloss && B > 0 && F > 0, "weighted_loss_rows: bad args");
    DAE_CHECK_ARG(loss_func >= DAE_LOSS_CROSS_ENTROPY && loss_func <= DAE_LOSS_COSINE, "weighted_loss_rows: unknown loss");
    DAE_LAUNCH(dae::weighted_loss_rows_kernel, dim3(B), dim3(256), 0, ST(stream), x, ldx, y, ldy, F, loss_func, rowloss);
    DAE_CHECK_LAUNCH();
    return 0;
}
