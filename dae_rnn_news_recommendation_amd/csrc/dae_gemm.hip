// dae_gemm.hip -- MFMA tile GEMM for the DAE hot path on gfx950 (MI355X).
//
//   C[M x N] = sum_seg A_seg[M x K_seg] * Bt_seg[N x K_seg]^T        ("NT": both operands K-contiguous)
//
// One kernel family serves every contraction of the training step (reference call sites:
// tf.sparse.matmul/tf.matmul autoencoder.py:389, tf.matmul :411, Gram triplet_loss_utils.py:93,219,
// and the three autodiff GEMMs of autoencoder.py:452-472):
//   encode   z1   = x~      . W         A = x~ [Bp x Fp]        Bt = W^T_lo [Hp x Fp]   split-K slabs
//   decode   z2   = h       . W^T       A = h  [Bp x Hp]        Bt = W_lo  [Fp x Hp]    fused loss epilogue
//   dh       dh   = delta2  . W  + Gs.h A = [delta2 | Gs]       Bt = [W^T_lo ; h^T]     split-K slabs
//   dW       dW   = x~^T.delta1 + delta2^T.h   A = [x~^T | delta2^T]  Bt = [delta1^T ; h^T]
//   gram     D    = h . h^T             exact fp32 MFMA
//
// Tiling (wave64, CDNA4): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators = 64 AGPR/VGPR), K-tile = 128 BYTES per row (64 bf16 /
// 32 fp32) so the LDS image, the swizzle and the fragment addressing are identical for both element
// types; only the MFMA differs:  bf16 -> 1 x v_mfma_f32_32x32x16_bf16 per 16-byte fragment,
// fp32 -> 4 x v_mfma_f32_32x32x2_f32 (exact fp32, the parity mode).
// LDS: 2 stages x (A 16 KiB + B 16 KiB) = 64 KiB -> 2 workgroups per CU.
// LDS image: row-major [128 rows][8 slots of 16 B], slot XOR-swizzled with (row>>1)&7 so that the
// 16 lanes of every ds_read_b128 lane group (rows distinct mod 16) hit 16 distinct 16-byte slots of
// the 256-byte bank row (conflict-free), and 8 consecutive lanes of the staging write cover one row.
// Staging: GLDS=true uses global_load_lds_dwordx4 (LDS image is lane-linear, so the swizzle is applied
// to the per-lane SOURCE address); GLDS=false stages through registers (global_load_dwordx4 ->
// ds_write_b128) with the loads issued before the MFMA block and the LDS write after it.
#include "dae_kernels.h"

#include <type_traits>

namespace dae {

constexpr int BM = 128, BN = 128;
constexpr int BKB = 128;                 // K-tile width in bytes
constexpr int GEMM_THREADS = 256;
constexpr int TILE_BYTES = BM * BKB;     // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int GEMM_LDS_BYTES = 2 * STAGE_BYTES;

struct GemmSeg {
    const char* A;
    const char* Bt;
    int64_t lda_b, ldb_b;   // leading dimensions in BYTES
    int ktiles;             // K_seg * sizeof(T) / 128
};

struct GemmParams {
    GemmSeg seg[2];
    int ktiles_total;
    int tiles_m, tiles_n, splits;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], c, 0, 0, 0);
    }
};

__device__ __forceinline__ void seg_of(const GemmParams& p, int kt, const char*& A, const char*& Bt,
                                       int64_t& lda, int64_t& ldb, int64_t& kbyte) {
    int s = (kt >= p.seg[0].ktiles) ? 1 : 0;
    int k = kt - (s ? p.seg[0].ktiles : 0);
    A = p.seg[s].A; Bt = p.seg[s].Bt; lda = p.seg[s].lda_b; ldb = p.seg[s].ldb_b;
    kbyte = (int64_t)k * BKB;
}

// ---- staging: register path ----
struct StageRegs { i32x4 a[4], b[4]; };

__device__ __forceinline__ void stage_load(const GemmParams& p, int kt, int row0_m, int row0_n, int tid, StageRegs& r) {
    const char *A, *Bt; int64_t lda, ldb, kb;
    seg_of(p, kt, A, Bt, lda, ldb, kb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + GEMM_THREADS * i;
        int row = c >> 3, slot = c & 7;
        r.a[i] = *reinterpret_cast<const i32x4*>(A + (int64_t)(row0_m + row) * lda + kb + slot * 16);
        r.b[i] = *reinterpret_cast<const i32x4*>(Bt + (int64_t)(row0_n + row) * ldb + kb + slot * 16);
    }
}
__device__ __forceinline__ void stage_write(char* stage, int tid, const StageRegs& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + GEMM_THREADS * i;
        int row = c >> 3, slot = c & 7;
        int off = row * BKB + ((slot ^ ((row >> 1) & 7)) << 4);
        *reinterpret_cast<i32x4*>(stage + off) = r.a[i];
        *reinterpret_cast<i32x4*>(stage + TILE_BYTES + off) = r.b[i];
    }
}

// ---- staging: direct global -> LDS (global_load_lds_dwordx4) ----
// wave w, piece i covers LDS bytes [(i*4+w)*1024, +1024) of each operand tile = 8 rows; lane l lands at
// +l*16, i.e. (row = (i*4+w)*8 + (l>>3), physical slot = l&7) and must fetch logical slot
// (l&7) ^ ((row>>1)&7) of that row.
__device__ __forceinline__ void stage_glds(const GemmParams& p, int kt, int row0_m, int row0_n, int wave, int lane,
                                           char* stage) {
    const char *A, *Bt; int64_t lda, ldb, kb;
    seg_of(p, kt, A, Bt, lda, ldb, kb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int piece = i * 4 + wave;
        int row = piece * 8 + (lane >> 3);
        int sslot = (lane & 7) ^ ((row >> 1) & 7);
        const char* ga = A + (int64_t)(row0_m + row) * lda + kb + sslot * 16;
        const char* gb = Bt + (int64_t)(row0_n + row) * ldb + kb + sslot * 16;
        char* la = stage + piece * 1024;
        char* lb = stage + TILE_BYTES + piece * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                         (__attribute__((address_space(3))) void*)la, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb,
                                         (__attribute__((address_space(3))) void*)lb, 16, 0, 0);
    }
}

template <typename T>
__device__ __forceinline__ void compute_stage(const char* stage, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const char* pa = stage + (wm * 64 + r) * BKB;
    const char* pb = stage + TILE_BYTES + (wn * 64 + r) * BKB;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int so = ((kk * 2 + g) ^ swz) << 4;
        i32x4 a0 = *reinterpret_cast<const i32x4*>(pa + so);
        i32x4 a1 = *reinterpret_cast<const i32x4*>(pa + 32 * BKB + so);
        i32x4 b0 = *reinterpret_cast<const i32x4*>(pb + so);
        i32x4 b1 = *reinterpret_cast<const i32x4*>(pb + 32 * BKB + so);
        Mma<T>::run(a0, b0, acc[0][0]);
        Mma<T>::run(a0, b1, acc[0][1]);
        Mma<T>::run(a1, b0, acc[1][0]);
        Mma<T>::run(a1, b1, acc[1][1]);
    }
}

template <typename T, bool GLDS>
__device__ __forceinline__ void gemm_mainloop(const GemmParams& p, int tm, int tn, int kt0, int kt1, char* lds,
                                              f32x16 (&acc)[2][2]) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int row0_m = tm * BM, row0_n = tn * BN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (kt0 >= kt1) return;

    if constexpr (GLDS) {
        stage_glds(p, kt0, row0_m, row0_n, wave, lane, lds);
        __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0) expcnt(0) lgkmcnt(0)
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            char* cur = lds + ((kt - kt0) & 1) * STAGE_BYTES;
            char* nxt = lds + (((kt - kt0) & 1) ^ 1) * STAGE_BYTES;
            if (kt + 1 < kt1) stage_glds(p, kt + 1, row0_m, row0_n, wave, lane, nxt);
            compute_stage<T>(cur, wm, wn, lane, acc);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
    } else {
        StageRegs regs;
        stage_load(p, kt0, row0_m, row0_n, tid, regs);
        stage_write(lds, tid, regs);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            char* cur = lds + ((kt - kt0) & 1) * STAGE_BYTES;
            char* nxt = lds + (((kt - kt0) & 1) ^ 1) * STAGE_BYTES;
            const bool more = (kt + 1 < kt1);
            if (more) stage_load(p, kt + 1, row0_m, row0_n, tid, regs);
            compute_stage<T>(cur, wm, wn, lane, acc);
            if (more) stage_write(nxt, tid, regs);
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void block_to_tile(const GemmParams& p, int& tm, int& tn, int& split, int& kt0, int& kt1) {
    const int id = blockIdx.x;
    split = id % p.splits;              // same K-slice -> same XCD when splits % 8 == 0 (block b -> XCD b%8)
    const int tile = id / p.splits;
    tn = tile % p.tiles_n;
    tm = tile / p.tiles_n;
    kt0 = (int)(((int64_t)p.ktiles_total * split) / p.splits);
    kt1 = (int)(((int64_t)p.ktiles_total * (split + 1)) / p.splits);
}

// ------------------------------------------------------------------------------------------------
// plain fp32-output kernel (split-K slabs or final C)
// ------------------------------------------------------------------------------------------------
template <typename T, bool GLDS>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_f32out(GemmParams p, float* __restrict__ C, int64_t ldc,
                                                                  int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int tm, tn, split, kt0, kt1;
    block_to_tile(p, tm, tn, split, kt0, kt1);
    f32x16 acc[2][2];
    gemm_mainloop<T, GLDS>(p, tm, tn, kt0, kt1, lds, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 5, c = lane & 31;
    float* Cs = C + (int64_t)split * slab_stride;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = tm * BM + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                int col = tn * BN + wn * 64 + nt * 32 + c;
                Cs[(int64_t)row * ldc + col] = acc[mt][nt][r];
            }
}

// ------------------------------------------------------------------------------------------------
// decode kernel: GEMM + bias + activation + reconstruction loss + d cost/d z2 (+ bias-gradient partials)
// autoencoder.py:411, triplet_loss_utils.py:262-277
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4*>(p) = v;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    uint2 v;
    v.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
    v.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
    *reinterpret_cast<uint2*>(p) = v;
}

__device__ __forceinline__ float half_sum32(float v) {   // sum over the 32 lanes sharing lane>>5
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, bool GLDS>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_decode_loss(GemmParams p, DecodeEpi e) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int tm, tn, split, kt0, kt1;
    block_to_tile(p, tm, tn, split, kt0, kt1);
    f32x16 acc[2][2];
    gemm_mainloop<T, GLDS>(p, tm, tn, kt0, kt1, lds, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 5, c = lane & 31;
    const T* X = reinterpret_cast<const T*>(e.x);
    T* D2 = reinterpret_cast<T*>(e.delta2);
    T* D2T = reinterpret_cast<T*>(e.delta2_t);
    const int colbase = tn * BN + wn * 64 + c;
    const int rowbase = tm * BM + wm * 64 + 4 * g;
    const float eps = 1e-16f;
    const bool is_cos = e.loss_func == DAE_LOSS_COSINE;

    float bvv[2];
    bool colok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int col = colbase + nt * 32;
        colok[nt] = col < e.F;
        bvv[nt] = colok[nt] ? e.bv[col] : 0.f;
    }
    float colsum[2] = {0.f, 0.f};

    // static (compile-time) accumulator indexing: a runtime-indexed f32x16 would be demoted to scratch
    auto epi_block = [&](auto MT, auto R4) {
        constexpr int mt = decltype(MT)::value, r4 = decltype(R4)::value;
            float d2v[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r4 * 4 + q;
                const int row = rowbase + mt * 32 + 8 * r4 + q;
                const bool rowok = row < e.B;
                const float cwi = e.cw[row];                  // zero beyond B by construction
                float rl = 0.f, s_yy = 0.f, s_xy = 0.f;
                // cosine_proximity (tf.nn.l2_normalize on both operands, triplet_loss_utils.py:273):
                // cos_stats = [sum x^2 | sum y^2 | sum xhat.y] per row; pass 1 produces the last two.
                float inx = 0.f, cs_yy = 0.f, cs_xy = 0.f;
                if (is_cos) {
                    inx = rsqrtf(fmaxf(e.cos_stats[row], 1e-12f));
                    if (e.cos_pass == 2) { cs_yy = e.cos_stats[e.Bp + row]; cs_xy = e.cos_stats[2 * e.Bp + row]; }
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int col = colbase + nt * 32;
                    const bool ok = rowok && colok[nt];
                    float z = acc[mt][nt][r] + bvv[nt];
                    float y = act_apply(e.dec_act, z);
                    float x = ok ? Elem<T>::to(X[(int64_t)row * e.ldx + col]) : 0.f;
                    float l = 0.f, dy = 0.f;
                    if (e.loss_func == DAE_LOSS_CROSS_ENTROPY) {
                        float a = y + eps, b = (1.0f - y) + eps;      // reference op order: (1.-y)+1e-16
                        l = -(x * __logf(a) + (1.0f - x) * __logf(b));
                        dy = -(x / a - (1.0f - x) / b);               // TF differentiates the two logs separately
                    } else if (e.loss_func == DAE_LOSS_MEAN_SQUARED) {
                        float d = x - y;
                        l = d * d;
                        dy = -2.0f * d;
                    } else {
                        float xh = x * inx;
                        if (e.cos_pass == 1) {
                            s_yy += ok ? y * y : 0.f;
                            s_xy += ok ? xh * y : 0.f;
                        } else {
                            float big = cs_yy >= 1e-12f ? 1.f : 0.f;  // tf.maximum routes grad to sum y^2 iff >= eps
                            float s = rsqrtf(fmaxf(cs_yy, 1e-12f));
                            dy = -(xh * s - big * cs_xy * s * s * s * y);
                        }
                    }
                    float d2 = (ok && e.cos_pass != 1) ? cwi * dy * act_grad(e.dec_act, y) : 0.f;
                    rl += ok ? l : 0.f;
                    d2v[nt][q] = d2;
                    colsum[nt] += d2;
                    if (e.y_out && ok) e.y_out[(int64_t)row * e.ldy + col] = y;
                    if (D2 && e.cos_pass != 1) D2[(int64_t)row * e.ldd + col] = Elem<T>::from(d2);
                }
                if (e.cos_pass == 1) {
                    s_yy = half_sum32(s_yy); s_xy = half_sum32(s_xy);
                    if (c == 0) {
                        int pw = tn * 2 + wn;
                        e.cos_part[(int64_t)pw * e.Bp + row] = s_yy;
                        e.cos_part[(int64_t)(2 * p.tiles_n + pw) * e.Bp + row] = s_xy;
                    }
                } else if (!is_cos) {
                    rl = half_sum32(rl);
                    if (c == 0) e.rowloss_part[(int64_t)(tn * 2 + wn) * e.Bp + row] = rl;
                }
            }
            if (D2T && e.cos_pass != 1) {
                const int row0 = rowbase + mt * 32 + 8 * r4;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int col = colbase + nt * 32;
                    store4<T>(D2T + (int64_t)col * e.lddt + row0, d2v[nt][0], d2v[nt][1], d2v[nt][2], d2v[nt][3]);
                }
            }
    };
#define DAE_EPI_ROWS(MTV)                                                                       \
    epi_block(std::integral_constant<int, MTV>{}, std::integral_constant<int, 0>{});            \
    epi_block(std::integral_constant<int, MTV>{}, std::integral_constant<int, 1>{});            \
    epi_block(std::integral_constant<int, MTV>{}, std::integral_constant<int, 2>{});            \
    epi_block(std::integral_constant<int, MTV>{}, std::integral_constant<int, 3>{});
    DAE_EPI_ROWS(0)
    DAE_EPI_ROWS(1)
#undef DAE_EPI_ROWS
    if (e.dbv_part && e.cos_pass != 1) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float v = colsum[nt] + __shfl_xor(colsum[nt], 32, 64);
            if (g == 0) e.dbv_part[(int64_t)(tm * 2 + wm) * e.Fp + colbase + nt * 32] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static bool g_use_glds = true;

static int fill_params(GemmParams& p, int dtype, int M, int N, const void* A0, int64_t lda0, const void* Bt0,
                       int64_t ldb0, int K0, const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int K1,
                       int splits) {
    const int es = (dtype == DAE_BF16) ? 2 : 4;
    const int kel = BKB / es;
    DAE_CHECK_ARG(dtype == DAE_BF16 || dtype == DAE_F32, "gemm: bad dtype %d", dtype);
    DAE_CHECK_ARG(M > 0 && N > 0 && M % BM == 0 && N % BN == 0, "gemm: M=%d N=%d must be positive multiples of 128", M, N);
    DAE_CHECK_ARG(K0 > 0 && K0 % kel == 0 && K1 >= 0 && K1 % kel == 0, "gemm: K0=%d K1=%d must be multiples of %d", K0, K1, kel);
    DAE_CHECK_ARG(A0 && Bt0 && (K1 == 0 || (A1 && Bt1)), "gemm: null operand");
    DAE_CHECK_ARG((lda0 * es) % 16 == 0 && (ldb0 * es) % 16 == 0 && (lda1 * es) % 16 == 0 && (ldb1 * es) % 16 == 0,
                  "gemm: leading dimensions must be 16-byte multiples");
    DAE_CHECK_ARG(((uintptr_t)A0 % 16) == 0 && ((uintptr_t)Bt0 % 16) == 0 && ((uintptr_t)A1 % 16) == 0 && ((uintptr_t)Bt1 % 16) == 0,
                  "gemm: operands must be 16-byte aligned");
    p.seg[0] = {(const char*)A0, (const char*)Bt0, lda0 * es, ldb0 * es, K0 / kel};
    p.seg[1] = {(const char*)A1, (const char*)Bt1, lda1 * es, ldb1 * es, K1 / kel};
    p.ktiles_total = p.seg[0].ktiles + p.seg[1].ktiles;
    p.tiles_m = M / BM; p.tiles_n = N / BN;
    p.splits = splits < 1 ? 1 : splits;
    DAE_CHECK_ARG(p.splits <= p.ktiles_total, "gemm: splits=%d exceeds k-tiles=%d", p.splits, p.ktiles_total);
    return 0;
}

template <typename K> static int set_lds(K kernel) {
    DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      GEMM_LDS_BYTES));
    return 0;
}
static int gemm_init() {
    static int rc = [] {
        if (int r = set_lds(gemm_nt_f32out<bf16_t, true>)) return r;
        if (int r = set_lds(gemm_nt_f32out<bf16_t, false>)) return r;
        if (int r = set_lds(gemm_nt_f32out<float, true>)) return r;
        if (int r = set_lds(gemm_nt_f32out<float, false>)) return r;
        if (int r = set_lds(gemm_decode_loss<bf16_t, true>)) return r;
        if (int r = set_lds(gemm_decode_loss<bf16_t, false>)) return r;
        if (int r = set_lds(gemm_decode_loss<float, true>)) return r;
        if (int r = set_lds(gemm_decode_loss<float, false>)) return r;
        return 0;
    }();
    return rc;
}

int launch_gemm_f32out(int dtype, int M, int N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0, int K0,
                       const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int K1, float* C, int64_t ldc,
                       int splits, int64_t slab_stride, hipStream_t st) {
    GemmParams p;
    if (int rc = fill_params(p, dtype, M, N, A0, lda0, Bt0, ldb0, K0, A1, lda1, Bt1, ldb1, K1, splits)) return rc;
    DAE_CHECK_ARG(C != nullptr, "gemm: C is null");
    if (int rc = gemm_init()) return rc;
    dim3 grid(p.tiles_m * p.tiles_n * p.splits), block(GEMM_THREADS);
#define DAE_LAUNCH_F32OUT(T, G)                                                              \
    do {                                                                                     \
        hipLaunchKernelGGL((gemm_nt_f32out<T, G>), grid, block, GEMM_LDS_BYTES, st, p, C, ldc, slab_stride); \
    } while (0)
    if (dtype == DAE_BF16) { if (g_use_glds) DAE_LAUNCH_F32OUT(bf16_t, true); else DAE_LAUNCH_F32OUT(bf16_t, false); }
    else                   { if (g_use_glds) DAE_LAUNCH_F32OUT(float, true);  else DAE_LAUNCH_F32OUT(float, false); }
#undef DAE_LAUNCH_F32OUT
    DAE_CHECK_LAUNCH();
    return 0;
}

int launch_decode_loss(int dtype, int Bp, int Fp, int Hp, const void* h_lo, int64_t ldh, const void* W_lo, int64_t ldw,
                       const DecodeEpi& e, hipStream_t st) {
    GemmParams p;
    if (int rc = fill_params(p, dtype, Bp, Fp, h_lo, ldh, W_lo, ldw, Hp, nullptr, 0, nullptr, 0, 0, 1)) return rc;
    if (int rc = gemm_init()) return rc;
    dim3 grid(p.tiles_m * p.tiles_n), block(GEMM_THREADS);
#define DAE_LAUNCH_DEC(T, G)                                                                 \
    do {                                                                                     \
        hipLaunchKernelGGL((gemm_decode_loss<T, G>), grid, block, GEMM_LDS_BYTES, st, p, e); \
    } while (0)
    if (dtype == DAE_BF16) { if (g_use_glds) DAE_LAUNCH_DEC(bf16_t, true); else DAE_LAUNCH_DEC(bf16_t, false); }
    else                   { if (g_use_glds) DAE_LAUNCH_DEC(float, true);  else DAE_LAUNCH_DEC(float, false); }
#undef DAE_LAUNCH_DEC
    DAE_CHECK_LAUNCH();
    return 0;
}

void set_use_glds(bool v) { g_use_glds = v; }

}  // namespace dae
