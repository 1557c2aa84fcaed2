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
//   gram     D    = h . h^T             exact fp32 MFMA, or split-bf16 (K = 3 Hp: [hi|hi|lo] . [hi|lo|hi]^T)
// A contraction walks up to GEMM_MAX_SEG = 6 K segments, each with its own operand pair and leading dimensions (seg_locate).  The
// split-bf16 precision mode (DAE_BF16X3) uses them for x = hi + lo operands: decode (h_hi,W_hi) (h_hi,W_lo) (h_lo,W_hi); dh and dW
// likewise, 5 segments each -- the kernels are the bf16 ones, only the segment lists and the lo images of the epilogues differ.
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
#include "dae_sym.h"
#include "dae_kernels.h"
#include "dae_label.h"

#include <type_traits>

namespace dae {

constexpr int BM = 128, BN = 128;
constexpr int BKB = 128;                 // K-tile width in bytes
constexpr int GEMM_THREADS = 256;
constexpr int TILE_BYTES = BM * BKB;     // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
// NST = number of LDS stages of the global_load_lds ring (0 = legacy register staging, 2 buffers)
constexpr int lds_bytes_for(int nst) { return (nst < 2 ? 2 : nst) * STAGE_BYTES; }
constexpr int wg_per_cu_for(int nst) { return nst <= 2 ? 2 : 1; }

struct GemmSeg {
    const char* A;
    const char* Bt;
    int64_t lda_b, ldb_b;   // leading dimensions in BYTES
    int ktiles;             // K_seg * sizeof(T) / 128
};

constexpr int GEMM_MAX_SEG = 6;    // K segments of one contraction: split-bf16 operands need (hi,hi) (hi,lo) (lo,hi) per product (dW with a valued x~^T: 2 x 3)
struct GemmParams {
    GemmSeg seg[GEMM_MAX_SEG];
    const char* bt2[GEMM_MAX_SEG];   // gemm_dw_pc<PAIR> only: a second Bt operand of the segment (same leading dimension) that shares its A tiles, or NULL
    int nseg;                  // non-empty segments, walked in order
    int epi_vec;               // gemm_nt_pc: 1 = LDS-staged epilogue (16-byte pieces, all 8 waves), 0 = dword stores from the accumulator layout (A/B)
    int ktiles_total;
    int tiles_m, tiles_n, splits;
    unsigned long long* trace;   // dae_gemm_trace only: [blocks][4 waves][8] shader-clock sums per K-loop phase
    float out_scale;           // fp32-output kernels: C = out_scale * accumulator (1 except for the dW gradient of scaled 16-bit delta images)
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& c) {
        // the 16-bit storage format of this build (dae_common.h): fp16 images multiply on v_mfma_f32_32x32x16_f16, bf16 images on ..._bf16 (same rate)
        if constexpr (kF16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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

// segment of K tile kt: its index, the tile's position inside it and the first K tile AFTER it (where the stream switches operands)
template <typename P>
__device__ __forceinline__ int seg_locate(const P& p, int kt, int& k_in_seg, int& seg_end) {
    int sg = 0, base = 0;
    while (sg + 1 < p.nseg && kt >= base + p.seg[sg].ktiles) { base += p.seg[sg].ktiles; ++sg; }
    k_in_seg = kt - base;
    seg_end = base + p.seg[sg].ktiles;
    return sg;
}
__device__ __forceinline__ void seg_of(const GemmParams& p, int kt, const char*& A, const char*& Bt,
                                       int64_t& lda, int64_t& ldb, int64_t& kbyte) {
    int k, end;
    const int s = seg_locate(p, kt, k, end);
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

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One K tile of MFMA work for a wave.  All 16 fragment reads (ds_read_b128, 64 VGPRs) are issued back to back and
// the four MFMA groups wait with COUNTED lgkmcnt (12/8/4/0): the first MFMAs start as soon as their fragments land
// and the LDS latency of the rest hides behind them.  rocprofv3 showed ~50 % of wave time parked in lgkmcnt(0) with
// the compiler's own read->wait(0)->MFMA x4 schedule, and hipcc turns any source-level hoisting back into a full
// wait, so the reads are inline asm (invisible to its scoreboard) with hand-placed waits; each wait is followed by
// sched_barrier(0) because register-only MFMAs may otherwise be hoisted above an asm s_waitcnt (guide 5.4 rule 18).
__device__ __forceinline__ i32x4 lds_read_b128(uint32_t addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ i32x4 lds_read_b128_off4096(uint32_t addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(v) : "v"(addr));
    return v;
}

template <typename T>
__device__ __forceinline__ void compute_stage(const char* stage, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)stage;
    const uint32_t pa = base + (wm * 64 + r) * BKB;
    const uint32_t pb = base + TILE_BYTES + (wn * 64 + r) * BKB;
    i32x4 a[4][2], b[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const uint32_t so = ((kk * 2 + g) ^ swz) << 4;
        a[kk][0] = lds_read_b128(pa + so);
        a[kk][1] = lds_read_b128_off4096(pa + so);          // + 32 rows * 128 B
        b[kk][0] = lds_read_b128(pb + so);
        b[kk][1] = lds_read_b128_off4096(pb + so);
    }
#define DAE_MMA_GROUP(KK, CNT)                                   \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<T>::run(a[KK][0], b[KK][0], acc[0][0]);                  \
    Mma<T>::run(a[KK][0], b[KK][1], acc[0][1]);                  \
    Mma<T>::run(a[KK][1], b[KK][0], acc[1][0]);                  \
    Mma<T>::run(a[KK][1], b[KK][1], acc[1][1]);
    DAE_MMA_GROUP(0, 12)
    DAE_MMA_GROUP(1, 8)
    DAE_MMA_GROUP(2, 4)
    DAE_MMA_GROUP(3, 0)
#undef DAE_MMA_GROUP
    __builtin_amdgcn_sched_barrier(0);
}

// K loop.  NST >= 2: ring of NST LDS stages filled by global_load_lds with COUNTED vmcnt waits -- tile i+NST-1 is
// requested right after the barrier of iteration i (its buffer was last read in iteration i-1), and the wait in
// front of the barrier only retires tile i, leaving up to NST-2 younger tiles (8 LDS-DMA ops per wave each) in
// flight across the barrier.  One raw s_barrier per K tile; __syncthreads() would drain the DMA queue (its
// fence waits vmcnt(0) while LDS-DMA writes are pending).
template <typename T, int NST, bool TRACE = false>
__device__ __forceinline__ void gemm_mainloop(const GemmParams& p, int tm, int tn, int kt0, int kt1, char* lds,
                                              f32x16 (&acc)[2][2]) {
    // TRACE (dae_gemm_trace): s_memtime stamps between the phases of every K iteration, summed per wave:
    //   [0] phase (a): 8 MFMAs (+DMA)  [1] waits (vmcnt, lgkmcnt)  [2] barrier  [3] phases (c,d,e): reads, 8 MFMAs + DMA, reads
    //   [4] iterations
    unsigned long long tsum[5] = {0, 0, 0, 0, 0}, tprev = 0;
#define DAE_STAMP(K)                                                         \
    if constexpr (TRACE) {                                                   \
        const unsigned long long t__ = __builtin_amdgcn_s_memtime();         \
        tsum[K] += t__ - tprev;                                              \
        tprev = t__;                                                         \
    }
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
    const int nk = kt1 - kt0;
    if (nk <= 0) return;

    if constexpr (NST >= 2) {
        // ---- LDS-DMA addressing: this lane's 16-byte chunk of each of its 4 pieces per operand, as a 32-bit byte offset
        //      from a uniform (SGPR) panel pointer that advances by one K tile per stage ----
        uint32_t voA[4], voB[4];
        const char *gA = nullptr, *gB = nullptr;
        int kt_dma = kt0, seg_end = 0;
        auto seg_setup = [&](int kt) {
            int k;
            const int sg = seg_locate(p, kt, k, seg_end);
            const uint32_t lda = (uint32_t)p.seg[sg].lda_b, ldb = (uint32_t)p.seg[sg].ldb_b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (i * 4 + wave) * 8 + (lane >> 3);
                const uint32_t ss = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
                voA[i] = (uint32_t)(row0_m + row) * lda + ss;
                voB[i] = (uint32_t)(row0_n + row) * ldb + ss;
            }
            gA = p.seg[sg].A + (int64_t)k * BKB;
            gB = p.seg[sg].Bt + (int64_t)k * BKB;
        };
        seg_setup(kt0);
        auto dma_piece = [&](int i, char* slot) {        // piece i of both operands of the stage at (gA, gB)
            const int piece = i * 4 + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                             (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + piece * 1024), 16, 0, 0);
        };
        auto dma_advance = [&]() {
            ++kt_dma;
            if (kt_dma == seg_end && kt_dma < p.ktiles_total) seg_setup(kt_dma);
            else { gA += BKB; gB += BKB; }
        };
        // Stage s lives in slot s % NST.  Rolling schedule of iteration i (fragment registers R0 = kk 0,1 and R1 = kk 2,3):
        //   (a) 8 MFMAs on R0(i)            [NST >= 3: + second half of the DMA of stage i+NST-1]
        //   (b) wait: stage i+1 landed, my LDS reads of tile i done; s_barrier
        //   (c) 8 ds_read_b128 of tile i+1 -> R0
        //   (d) 8 MFMAs on R1(i)            + DMA of stage i+NST into slot i % NST (first half when NST >= 3)
        //   (e) 8 ds_read_b128 of tile i+1 -> R1
        // so the fragment reads of the next tile and the LDS-DMA issue run under the MFMAs of this tile; the only exposed
        // latency per K tile is the barrier.  Reads past the last tile fetch stale LDS and are never consumed.
        constexpr bool SPLIT = NST >= 3;
        const int r = lane & 31, g = lane >> 5;
        const int swz = (r >> 1) & 7;
        const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
        const uint32_t offa = (wm * 64 + r) * BKB, offb = TILE_BYTES + (wn * 64 + r) * BKB;
        uint32_t so[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
        i32x4 fa[4][2], fb[4][2];
#define DAE_READ_KK(KK, SLOTBASE)                                          \
    fa[KK][0] = lds_read_b128((SLOTBASE) + offa + so[KK]);                 \
    fa[KK][1] = lds_read_b128_off4096((SLOTBASE) + offa + so[KK]);         \
    fb[KK][0] = lds_read_b128((SLOTBASE) + offb + so[KK]);                 \
    fb[KK][1] = lds_read_b128_off4096((SLOTBASE) + offb + so[KK]);
#define DAE_MMA2(KK, MT)                                                   \
    Mma<T>::run(fa[KK][MT], fb[KK][0], acc[MT][0]);                        \
    Mma<T>::run(fa[KK][MT], fb[KK][1], acc[MT][1]);                        \
    __builtin_amdgcn_sched_barrier(0);

        // ---- prologue: request stages 0..NST-2 (+ first half of NST-1 when SPLIT, else all of NST-1) ----
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            if (st < nk) {
                char* slot = lds + st * STAGE_BYTES;
                if (SPLIT && st == NST - 1) { dma_piece(0, slot); dma_piece(1, slot); }
                else { dma_piece(0, slot); dma_piece(1, slot); dma_piece(2, slot); dma_piece(3, slot); dma_advance(); }
            }
        }
        if (nk >= NST) { if constexpr (SPLIT) wait_vm<(NST - 2) * 8 + 4>(); else wait_vm<(NST - 1) * 8>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        DAE_READ_KK(0, lbase) DAE_READ_KK(1, lbase) DAE_READ_KK(2, lbase) DAE_READ_KK(3, lbase)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TRACE) tprev = __builtin_amdgcn_s_memtime();
        int cur = 0;                                      // slot of tile i
        for (int i = 0; i < nk; ++i) {
            const int nxt = cur + 1 == NST ? 0 : cur + 1;
            // (a)
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            DAE_MMA2(0, 0)
            if constexpr (SPLIT) { if (i + NST - 1 < nk) dma_piece(2, lds + (cur == 0 ? NST - 1 : cur - 1) * STAGE_BYTES); __builtin_amdgcn_sched_barrier(0); }
            DAE_MMA2(0, 1)
            DAE_MMA2(1, 0)
            if constexpr (SPLIT) { if (i + NST - 1 < nk) { dma_piece(3, lds + (cur == 0 ? NST - 1 : cur - 1) * STAGE_BYTES); dma_advance(); } __builtin_amdgcn_sched_barrier(0); }
            DAE_MMA2(1, 1)
            DAE_STAMP(0)
            // (b)
            {
                const int ahead = min(NST - 2, nk - 2 - i);
                if (ahead >= 2) wait_vm<16>();
                else if (ahead == 1) wait_vm<8>();
                else wait_vm<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DAE_STAMP(1)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            DAE_STAMP(2)
            // (c)
            {
                const uint32_t nb = lbase + nxt * STAGE_BYTES;
                DAE_READ_KK(0, nb) DAE_READ_KK(1, nb)
                __builtin_amdgcn_sched_barrier(0);
                // (d): R1 landed at (b)
                char* slot = lds + cur * STAGE_BYTES;
                const bool more = i + NST < nk;
                DAE_MMA2(2, 0)
                if (more) dma_piece(0, slot);
                __builtin_amdgcn_sched_barrier(0);
                DAE_MMA2(2, 1)
                if (more) dma_piece(1, slot);
                __builtin_amdgcn_sched_barrier(0);
                DAE_MMA2(3, 0)
                if constexpr (!SPLIT) { if (more) dma_piece(2, slot); __builtin_amdgcn_sched_barrier(0); }
                DAE_MMA2(3, 1)
                if constexpr (!SPLIT) { if (more) { dma_piece(3, slot); dma_advance(); } __builtin_amdgcn_sched_barrier(0); }
                // (e)
                DAE_READ_KK(2, nb) DAE_READ_KK(3, nb)
                __builtin_amdgcn_sched_barrier(0);
            }
            DAE_STAMP(3)
            cur = nxt;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // drain the stale tail reads before the LDS is reused
#undef DAE_READ_KK
#undef DAE_MMA2
        if constexpr (TRACE) {
            if (lane == 0) {
                unsigned long long* o = p.trace + ((size_t)blockIdx.x * 4 + wave) * 8;
                o[0] = tsum[0]; o[1] = tsum[1]; o[2] = tsum[2]; o[3] = tsum[3]; o[4] = (unsigned long long)nk;
            }
        }
#undef DAE_STAMP
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

// block -> (tile, K slice).  The dispatcher places block b on XCD b % 8 (each XCD has a private 4 MiB L2), so:
//   splits > 1 : split = b % splits -- every block of one K slice lands on the same XCD (splits % 8 == 0) and
//                the slice of both operands (a few MiB) is fetched from HBM once per XCD;
//   splits == 1: the longer tile dimension is cut into 8 contiguous bands, one per XCD, so a band's operand panel
//                is read from HBM by ONE XCD and re-used from its L2 by the tiles that share it (rocprofv3
//                FETCH_SIZE of the dW GEMM: 145 MB with the naive row-major map vs 38 MB of operands).
// Placement only affects speed, never results.  Returns false for the few padding blocks of the banded map.
__device__ __forceinline__ bool block_to_tile(const GemmParams& p, int& tm, int& tn, int& split, int& kt0, int& kt1) {
    const int id = blockIdx.x;
    if (p.splits > 1) {
        split = id % p.splits;
        const int tile = id / p.splits;
        tn = tile % p.tiles_n;
        tm = tile / p.tiles_n;
    } else {
        split = 0;
        const int x = id & 7, l = id >> 3;
        if (p.tiles_m >= p.tiles_n) {
            const int per = (p.tiles_m + 7) >> 3;
            tm = x * per + l / p.tiles_n;
            tn = l % p.tiles_n;
            if (l >= per * p.tiles_n || tm >= p.tiles_m) return false;
        } else {
            const int per = (p.tiles_n + 7) >> 3;
            tn = x * per + l / p.tiles_m;
            tm = l % p.tiles_m;
            if (l >= per * p.tiles_m || tn >= p.tiles_n) return false;
        }
    }
    kt0 = (int)(((int64_t)p.ktiles_total * split) / p.splits);
    kt1 = (int)(((int64_t)p.ktiles_total * (split + 1)) / p.splits);
    return true;
}
static int grid_blocks(const GemmParams& p) {
    if (p.splits > 1) return p.tiles_m * p.tiles_n * p.splits;
    const int big = p.tiles_m >= p.tiles_n ? p.tiles_m : p.tiles_n, small = p.tiles_m >= p.tiles_n ? p.tiles_n : p.tiles_m;
    return 8 * ((big + 7) / 8) * small;
}

}  // namespace dae
#include "dae_gemm_w8.h"      // 256 x 256 tiles, 8 MFMA waves: the large split-K contractions (uses GemmSeg / Mma / wait_vm above)
namespace dae {

// ------------------------------------------------------------------------------------------------
// plain fp32-output kernel (split-K slabs or final C)
// ------------------------------------------------------------------------------------------------
// ROLE only names the instantiation (encode / dh / dW / gram / generic) so that per-kernel profiles
// (rocprofv3 --kernel-trace) can tell the step's GEMMs apart; the code is identical.
enum { ROLE_GENERIC = 0, ROLE_ENCODE = 1, ROLE_DH = 2, ROLE_DW = 3, ROLE_GRAM = 4 };
template <typename T, int NST, int ROLE>
__global__ __launch_bounds__(GEMM_THREADS, wg_per_cu_for(NST)) void gemm_nt_f32out(GemmParams p, float* __restrict__ C, int64_t ldc,
                                                                                   int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int tm, tn, split, kt0, kt1;
    if (!block_to_tile(p, tm, tn, split, kt0, kt1)) return;
    f32x16 acc[2][2];
    gemm_mainloop<T, NST>(p, tm, tn, kt0, kt1, lds, acc);
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
                Cs[(int64_t)row * ldc + col] = acc[mt][nt][r] * p.out_scale;
            }
}

// ------------------------------------------------------------------------------------------------
// dW GEMM with the optimizer fused into its epilogue (single-GPU step, bf16 mode):
//   g = x~^T delta1 + delta2^T h  (autoencoder.py:444-477: opt.minimize(cost) -> the W gradient and its apply op)
// the 128x128 gradient tile never leaves the registers: W (fp32 master), the optimizer slots and both bf16 shadow
// layouts (W_lo [F x H] for decode, Wt_lo [H x F] for encode / dh) are updated in place -- saves the 20 MB gradient
// round trip through HBM and one kernel.  The shadows leave the CU as coalesced 16-byte rows staged through LDS
// (same staging as the decode epilogue).  `grad` is still written when the caller wants to read it (NULL skips it).
// ------------------------------------------------------------------------------------------------
constexpr int DWO_PITCH = 272;                        // staged bf16 row: 128 elements + 16 B pad
constexpr int DWO_TILE_BYTES = 128 * DWO_PITCH;
constexpr int DWO_G_BYTES = 64 * 128 * 4;             // half of the gradient tile, fp32 [64][128]
constexpr int DWO_LDS = DWO_G_BYTES + DWO_TILE_BYTES; // 66 KiB (the two K-loop stages, 64 KiB, are dead by then)

// Epilogue in two halves of 64 rows: the two waves that own the half park their accumulators in LDS (fp32, conflict-free from the
// accumulator layout), then all 4 waves run the optimizer on blocks of 4 rows x 4 columns -- every global access is a 16-byte piece of a
// 512-byte row run, a quarter of the instructions of the per-lane form (same scheme as gemm_dw_pc, where it was measured).  W_lo
// leaves as 8-byte pieces; the transposed shadow is staged in LDS and leaves as 16-byte pieces at the end.
template <int OPT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_dw_opt(GemmParams p, OptEpi e) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int tm, tn, split, kt0, kt1;
    if (!block_to_tile(p, tm, tn, split, kt0, kt1)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 5, c = lane & 31;
    float* __restrict__ Wp = e.W;
    float* __restrict__ gradp = e.grad;
    float* __restrict__ s1p = e.s1;
    float* __restrict__ s2p = e.s2;
    const float lr = e.lr, mom = e.mom, gscale = e.gscale, gin = e.gin == 0.f ? 1.f : e.gin;
    // block (half, i) of this thread: rows half * 64 + 4 * rg .. + 3, columns 4 * c4 .. + 3 of the tile
    const int rg0 = tid >> 5, c4 = tid & 31;                                  // i-th block: row group rg0 + 8 i
    // plain SGD: this thread's 16 master-weight pieces are requested BEFORE the K loop and ride under it (the stateful optimizers load W
    // together with their slots, per block)
    constexpr bool PREFETCH_W = (OPT == DAE_OPT_SGD);
    f32x4 wq[2][2][4];
    if constexpr (PREFETCH_W) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    wq[h][i][q] = *reinterpret_cast<const f32x4*>(Wp + (int64_t)(tm * BM + h * 64 + (rg0 + 8 * i) * 4 + q) * e.ldw + tn * BN + c4 * 4);
    }
    f32x16 acc[2][2];
    gemm_mainloop<bf16_t, 2>(p, tm, tn, kt0, kt1, lds, acc);
    float* Gt = reinterpret_cast<float*>(lds);          // gradient half [64][128]
    char* R1 = lds + DWO_G_BYTES;                       // Wt_lo tile  [h_local][f_local]
    bf16_t* __restrict__ Wlo = reinterpret_cast<bf16_t*>(e.W_lo);
    auto half = [&](auto HV) {
        constexpr int h = decltype(HV)::value;
        __syncthreads();                                // the K-loop stages (h = 0) / the previous half's gradient reads are done
        if (wm == h) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Gt[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) * 128 + wn * 64 + nt * 32 + c] = acc[mt][nt][r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rg = rg0 + 8 * i;
            float pv[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lrow = rg * 4 + q;
                const int64_t k = (int64_t)(tm * BM + h * 64 + lrow) * e.ldw + tn * BN + c4 * 4;
                const f32x4 gr = *reinterpret_cast<const f32x4*>(Gt + lrow * 128 + c4 * 4) * gin;
                f32x4 p0, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, pn;
                if constexpr (PREFETCH_W) p0 = wq[h][i][q];
                else p0 = *reinterpret_cast<const f32x4*>(Wp + k);
                if constexpr (OPT != DAE_OPT_SGD) a1 = *reinterpret_cast<const f32x4*>(s1p + k);
                if constexpr (OPT == DAE_OPT_ADAM) a2 = *reinterpret_cast<const f32x4*>(s2p + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gg = gr[j] * gscale;
                    if constexpr (OPT == DAE_OPT_SGD) pn[j] = p0[j] - lr * gg;
                    else if constexpr (OPT == DAE_OPT_ADAGRAD) { const float a = a1[j] + gg * gg; a1[j] = a; pn[j] = p0[j] - lr * gg * rsqrtf(a); }
                    else if constexpr (OPT == DAE_OPT_MOMENTUM) { const float a = mom * a1[j] + gg; a1[j] = a; pn[j] = p0[j] - lr * a; }
                    else {
                        const float m = 0.9f * a1[j] + 0.1f * gg;
                        const float v = 0.999f * a2[j] + 0.001f * gg * gg;
                        a1[j] = m; a2[j] = v;
                        pn[j] = p0[j] - lr * m / (sqrtf(v) + 1e-8f);
                    }
                    pv[q][j] = pn[j];
                }
                *reinterpret_cast<f32x4*>(Wp + k) = pn;
                if (gradp) *reinterpret_cast<f32x4*>(gradp + k) = gr;
                if constexpr (OPT != DAE_OPT_SGD) *reinterpret_cast<f32x4*>(s1p + k) = a1;
                if constexpr (OPT == DAE_OPT_ADAM) *reinterpret_cast<f32x4*>(s2p + k) = a2;
                uint2 lo;
                lo.x = f2bf_pack_hw(pn[0], pn[1]); lo.y = f2bf_pack_hw(pn[2], pn[3]);
                *reinterpret_cast<uint2*>(Wlo + k) = lo;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                // 4 features of column 4 c4 + j: one 8-byte piece of the transposed tile
                uint2 v;
                v.x = f2bf_pack_hw(pv[0][j], pv[1][j]);
                v.y = f2bf_pack_hw(pv[2][j], pv[3][j]);
                *reinterpret_cast<uint2*>(R1 + (c4 * 4 + j) * DWO_PITCH + (h * 64 + rg * 4) * 2) = v;
            }
        }
    };
    half(std::integral_constant<int, 0>{});
    half(std::integral_constant<int, 1>{});
    __syncthreads();
    bf16_t* Wtlo = reinterpret_cast<bf16_t*>(e.Wt_lo);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = tid + GEMM_THREADS * i;
        const int row = ch >> 4, c16 = ch & 15;
        *reinterpret_cast<i32x4*>(Wtlo + (int64_t)(tn * BN + row) * e.ldwt + tm * BM + c16 * 8) =
            *reinterpret_cast<const i32x4*>(R1 + row * DWO_PITCH + c16 * 16);
    }
}

// ------------------------------------------------------------------------------------------------
// Producer/consumer variant for grids of at most one workgroup per CU (the split-K GEMMs: encode, dh, Gram).
// 8 waves: waves 0-3 compute (rolling fragment schedule of gemm_mainloop, no DMA instructions), waves 4-7 only feed the
// NST-slot LDS ring with global_load_lds.  Issuing one 1-KiB LDS-DMA piece costs the issuing wave ~60 cycles
// (gemm_trace: 8 pieces per K tile = as long as the tile's 16 MFMAs), so in the 4-wave kernel the matrix pipe idles
// during the DMA issue; here the second wave of each SIMD pays that cost.  One s_barrier per K tile for all 8 waves:
//   BARRIER_i : consumers have read tile i out of LDS (slot i % NST is free)  AND  producers have seen stage i+1 land.
// ------------------------------------------------------------------------------------------------
constexpr int PC_THREADS = 512;
constexpr int PC_NST = 4;                            // 128 KiB of the CU's 160 KiB LDS

template <typename T, int NST, int ROLE>
__global__ __launch_bounds__(PC_THREADS, 1) void gemm_nt_pc(GemmParams p, float* __restrict__ C, int64_t ldc, int64_t slab_stride,
                                                            LabelJob job, int label_block) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // the grid leaves CUs idle (28 x 8 = 224 workgroups of the encode GEMM on 256 CUs): one extra workgroup computes the
    // batch's label statistics (dae_label.h) there, hidden under the GEMM instead of lengthening another launch
    if ((int)blockIdx.x == label_block) { label_stats_block<PC_THREADS>(job, lds); return; }
    int tm, tn, split, kt0, kt1;
    if (!block_to_tile(p, tm, tn, split, kt0, kt1)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = kt1 - kt0;
    const int row0_m = tm * BM, row0_n = tn * BN;

    f32x16 acc[2][2];
    if (wave8 >= 4) {
        // ================= producer =================
        if (nk > 0) {
        const int wave = wave8 - 4;
        uint32_t voA[4], voB[4];
        const char *gA = nullptr, *gB = nullptr;
        int kt_dma = kt0, seg_end = 0;
        auto seg_setup = [&](int kt) {
            int k;
            const int sg = seg_locate(p, kt, k, seg_end);
            const uint32_t lda = (uint32_t)p.seg[sg].lda_b, ldb = (uint32_t)p.seg[sg].ldb_b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (i * 4 + wave) * 8 + (lane >> 3);
                const uint32_t ss = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
                voA[i] = (uint32_t)(row0_m + row) * lda + ss;
                voB[i] = (uint32_t)(row0_n + row) * ldb + ss;
            }
            gA = p.seg[sg].A + (int64_t)k * BKB;
            gB = p.seg[sg].Bt + (int64_t)k * BKB;
        };
        seg_setup(kt0);
        auto dma_stage = [&](char* slot) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = i * 4 + wave;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                                 (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + voB[i]),
                                                 (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + piece * 1024), 16, 0, 0);
            }
            ++kt_dma;
            if (kt_dma == seg_end && kt_dma < p.ktiles_total) seg_setup(kt_dma);
            else { gA += BKB; gB += BKB; }
        };
#pragma unroll
        for (int st = 0; st < NST; ++st)
            if (st < nk) dma_stage(lds + st * STAGE_BYTES);
        if (nk >= NST) wait_vm<(NST - 1) * 8>(); else wait_vm<0>();      // stage 0 landed
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int i = 0; i < nk; ++i) {
            const int ahead = min(NST - 2, nk - 2 - i);                   // stages younger than i+1 already requested
            if (ahead >= 2) wait_vm<16>();
            else if (ahead == 1) wait_vm<8>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (i + NST < nk) dma_stage(lds + cur * STAGE_BYTES);
            cur = cur + 1 == NST ? 0 : cur + 1;
        }
        }
    } else {
    // ================= consumer =================
    const int wave = wave8;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (nk > 0) {
        const int r = lane & 31, g = lane >> 5;
        const int swz = (r >> 1) & 7;
        const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
        const uint32_t offa = (wm * 64 + r) * BKB, offb = TILE_BYTES + (wn * 64 + r) * BKB;
        uint32_t so[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
        i32x4 fa[4][2], fb[4][2];
#define DAE_READ_KK(KK, SLOTBASE)                                          \
    fa[KK][0] = lds_read_b128((SLOTBASE) + offa + so[KK]);                 \
    fa[KK][1] = lds_read_b128_off4096((SLOTBASE) + offa + so[KK]);         \
    fb[KK][0] = lds_read_b128((SLOTBASE) + offb + so[KK]);                 \
    fb[KK][1] = lds_read_b128_off4096((SLOTBASE) + offb + so[KK]);
#define DAE_MMA4(KK)                                                       \
    Mma<T>::run(fa[KK][0], fb[KK][0], acc[0][0]);                          \
    Mma<T>::run(fa[KK][0], fb[KK][1], acc[0][1]);                          \
    Mma<T>::run(fa[KK][1], fb[KK][0], acc[1][0]);                          \
    Mma<T>::run(fa[KK][1], fb[KK][1], acc[1][1]);
        __builtin_amdgcn_s_barrier();                                     // stage 0 landed (producers waited for it)
        asm volatile("" ::: "memory");
        DAE_READ_KK(0, lbase) DAE_READ_KK(1, lbase) DAE_READ_KK(2, lbase) DAE_READ_KK(3, lbase)
        __builtin_amdgcn_sched_barrier(0);
        int cur = 0;
        for (int i = 0; i < nk; ++i) {
            const int nxt = cur + 1 == NST ? 0 : cur + 1;
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");            // R0 (kk 0,1) of tile i
            __builtin_amdgcn_sched_barrier(0);
            DAE_MMA4(0) DAE_MMA4(1)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // R1 landed; every LDS read of tile i is done
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint32_t nb = lbase + nxt * STAGE_BYTES;
            DAE_READ_KK(0, nb) DAE_READ_KK(1, nb)                         // stale (never consumed) after the last tile
            __builtin_amdgcn_sched_barrier(0);
            DAE_MMA4(2) DAE_MMA4(3)
            __builtin_amdgcn_sched_barrier(0);
            DAE_READ_KK(2, nb) DAE_READ_KK(3, nb)
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef DAE_READ_KK
#undef DAE_MMA4
    }
    }
    // ---- epilogue, all 8 waves: the accumulators are parked in LDS (fp32 [128][128], 64 KiB of the dead ring; conflict-free from the accumulator
    //      layout) and leave as 16-byte pieces of 512-byte row runs, 8 per thread.  (Before: 64 dword stores per lane from the four MFMA waves alone,
    //      two 128-byte runs per instruction, the producer waves gone -- the slab store tail was a third of the Gram launch.)
    if (!p.epi_vec) {                                                     // A/B (dae_set_glds(-9)): the former epilogue
        if (wave8 < 4) {
            const int wm = wave8 >> 1, wn = wave8 & 1, g = lane >> 5, c = lane & 31;
            float* Cd = C + (int64_t)split * slab_stride;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cd[(int64_t)(tm * BM + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) * ldc + tn * BN + wn * 64 + nt * 32 + c] = acc[mt][nt][r] * p.out_scale;
        }
        return;
    }
    __builtin_amdgcn_s_barrier();                                         // every wave is out of the K loop: the ring is dead (all LDS-DMA landed)
    asm volatile("" ::: "memory");
    float* Ct = reinterpret_cast<float*>(lds);
    if (wave8 < 4) {
        const int wm = wave8 >> 1, wn = wave8 & 1, g = lane >> 5, c = lane & 31;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ct[(wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g) * BN + wn * 64 + nt * 32 + c] = acc[mt][nt][r] * p.out_scale;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // the tile is complete
    asm volatile("" ::: "memory");
    float* Cs = C + (int64_t)split * slab_stride + (int64_t)(tm * BM) * ldc + tn * BN;
#pragma unroll
    for (int i = 0; i < (BM * BN / 4) / PC_THREADS; ++i) {
        const int q = tid + PC_THREADS * i, row = q >> 5, c4 = q & 31;
        *reinterpret_cast<f32x4*>(Cs + (int64_t)row * ldc + c4 * 4) = *reinterpret_cast<const f32x4*>(Ct + row * BN + c4 * 4);
    }
}

// ------------------------------------------------------------------------------------------------
// dW GEMM + optimizer as an 8-wave producer/consumer kernel on 160 x 128 tiles.
// gemm_dw_opt above runs 79 x 4 = 316 tiles of 128 x 128 with four waves that both feed the LDS ring and issue the MFMAs
// (K loop at ~30 % of the MFMA rate) on 2 workgroups per CU -- 316 tiles leave most CUs with ONE workgroup, so nothing hides
// the DMA issue.  Here the 10112 x 512 gradient is cut into 64 x 4 = 256 tiles of 160 x 128 -- exactly one per CU, one
// round -- and each workgroup is specialised like gemm_nt_pc: waves 4-7 only issue LDS-DMA (9 pieces of 1 KiB per wave and
// K tile into a 4-slot ring), waves 0-3 only read fragments and issue MFMAs.  Consumer wave w owns all 160 rows x columns
// [32 w, +32): 5 accumulators, per 16-deep k step 5 A fragments + 1 B fragment for 5 MFMAs; fragments are double buffered
// one k step ahead (48 VGPRs) so that the master weights of the tile can stay prefetched in registers (80 VGPRs, SGD).
// The last row tile is partial (10112 = 63 x 160 + 32): its out-of-range rows load row Fp-1 and are never stored.
//
// XBITS (binary CSR input): x~^T never exists as an 18 MB image that is mostly zeros.  It arrives as a BIT image (bit i of row f
// <=> entry (i, f) of the batch was kept; written by the encode launch, 1.1 MB) and the A tiles of the x~^T . delta1 segment
// are BUILT in LDS by the producer waves (zero fill + one 2-byte store per set bit, or an arithmetic expansion when a word is
// dense -- the popular features of a Zipf vocabulary are kept in most rows, so their rows are NOT sparse), exactly as
// gemm_encode_bits_pc builds x~.  Only delta1^T is streamed for that segment: 187 MB through the LDS-DMA path per launch instead
// of 258 MB, and an XCD's share of the operands (3.2 MB) fits its 4 MiB L2.  Same MFMA products and accumulation order as the
// dense image -> bit-identical gradients.  (A sum over the kept entries instead -- "sparse x~^T.delta1" -- was built and
// measured first: 0.11 GFLOP, but half of the entries sit in 3 % of the feature rows and one tile took 120 us;
// profiles/r03_experiments.md.)
// OPT == DW_GRAD_ONLY: no optimizer, the gradient tile goes to memory (fp32 `grad` and / or bf16 `grad_lo`): the data-parallel step.
// ------------------------------------------------------------------------------------------------
constexpr int DW_BM = 160, DW_MB = DW_BM / 32;                         // rows per tile, MFMA row blocks per consumer wave
constexpr int DW_A_BYTES = DW_BM * BKB;                                // 20 KiB
constexpr int DW_STAGE = DW_A_BYTES + TILE_BYTES;                      // + 16 KiB B tile = 36 KiB
constexpr int DW_NST = 4;
constexpr int DW_P0 = 128 * 2 + 16;                                    // staged W_lo row [160][128 bf16 + pad]
constexpr int DW_P1 = DW_BM * 2 + 16;                                  // staged Wt_lo row [128][160 bf16 + pad]
constexpr int DW_RING = DW_NST * DW_STAGE;                             // 144 KiB (the epilogue tiles, 86 KiB, reuse it)
constexpr int DW_GRAD_ONLY = DW_OPT_GRAD_ONLY;                         // OPT value: gradient to memory, no update
constexpr int DW_LDS = DW_RING;
constexpr int DWB_MAXKT = 16;                                          // XBITS: K tiles of the x~^T segment (Bp <= 1024)
// PAIR (split-bf16 mode): a stage holds ONE A tile and TWO B tiles (A . [B_hi ; B_lo]: the hi and lo images of delta1^T resp. h^T share the
// x~^T resp. delta2^T_hi tile), so the pair costs one A stream, 7 fragment reads per 10 MFMAs instead of 12, and one barrier instead of two
constexpr int DW_STAGE2 = DW_A_BYTES + 2 * TILE_BYTES;                 // 52 KiB
constexpr int DW_NST2 = 3;
constexpr int DW_RING2 = DW_NST2 * DW_STAGE2;                          // 156 KiB

struct DwBits {
    const uint32_t* xtb; int64_t ldxt;       // x~^T bit image [Mrows x ldxt words]
    int nwords;                              // Bp / 32
    uint32_t one;                            // bf16 bits of the value of a kept entry (the corruption's scale factor; 1.0 for masking noise)
};

__device__ __forceinline__ void wait_vm_n(int n) {   // counted vmcnt wait for the op counts a mixed (4 / 9 pieces per stage) ring can leave in flight
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 4: wait_vm<4>(); break;
        case 8: wait_vm<8>(); break;
        case 9: wait_vm<9>(); break;
        case 12: wait_vm<12>(); break;
        case 13: wait_vm<13>(); break;
        case 17: wait_vm<17>(); break;
        case 18: wait_vm<18>(); break;
        case 22: wait_vm<22>(); break;
        case 26: wait_vm<26>(); break;
        case 27: wait_vm<27>(); break;
        default: wait_vm<0>(); break;                    // an op count nobody planned for: drain (always correct)
    }
}

// TRA (round 5): the A operands are the ROW-MAJOR batch images x~ [Bp x Fp] and delta2 [Bp x Fp] -- K (the batch) is their ROW index -- instead of the
// transposed images x~^T / delta2^T, so the decode kernel stores delta2 once and the gathers write no x~^T.  An LDS stage then holds the A tile as
// [64 k][160 m] 16-bit (pitch 320 B: the same 20 KiB, filled by the same 20 LDS-DMA pieces, each lane fetching 8 consecutive features of one batch row) and
// the consumers read their MFMA A fragments with gfx950's transposing LDS read: lane (row m, k group g) gets A[m][8 g .. 8 g + 7] from TWO
// ds_read_b64_tr_b16 (4 k each; lane i of a 16-lane group addresses k row i >> 2, features 4 (i & 3) .. + 3; tools/tr_probe.hip pins the mapping on the
// box).  The pitch of 320 B puts the four k rows of a 32-lane access into disjoint bank groups (80 dwords = 16 mod 64): conflict-free.
typedef int i32x2 __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ i32x2 lds_read_tr16_b64(uint32_t addr) {
    i32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF));
    return v;
}
constexpr int DW_TR_PITCH = DW_BM * 2;                                  // bytes per k row of the [k][m] A image
static_assert(64 * DW_TR_PITCH == DW_A_BYTES, "the [k][m] image of a K tile fills the A tile exactly");

template <int OPT, bool XBITS, bool X3 = false, bool PAIR = false, bool TRA = false>      // X3 (split-bf16 mode): the lo images of both shadows are written too (e.W_lo2 / e.Wt_lo2)
__global__ __launch_bounds__(PC_THREADS, 1) void gemm_dw_pc(GemmParams p, OptEpi e, int Mrows, DwBits xb) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    static_assert(!PAIR || (X3 && !XBITS), "paired stages exist for the split-bf16 contraction on dense operand images");
    static_assert(!TRA || (!PAIR && !XBITS), "the transposed-A form streams plain K segments of dense row-major images");
    constexpr int STG = PAIR ? DW_STAGE2 : DW_STAGE;                        // bytes per ring stage
    constexpr int NSTG = PAIR ? DW_NST2 : DW_NST;                           // ring depth
    // PAIR: does K tile t belong to a segment with a second B operand?  (uniform; a walk over <= 6 segments)
    // (segment ends and pair flags are read from the kernel arguments ONCE, with static indices: the K loops call pair_of() every iteration and a
    // scalar load there would put an s_waitcnt lgkmcnt(0) in front of the counted fragment waits)
    int seg_end_[GEMM_MAX_SEG];
    uint32_t pairbits = 0u;
    if constexpr (PAIR) {
        int base = 0;
#pragma unroll
        for (int sgi = 0; sgi < GEMM_MAX_SEG; ++sgi) {
            base += sgi < p.nseg ? p.seg[sgi].ktiles : 0;
            seg_end_[sgi] = base;
            if (sgi < p.nseg && p.bt2[sgi] != nullptr) pairbits |= 1u << sgi;
        }
    }
    auto pair_of = [&](int t) -> bool {
        if constexpr (!PAIR) return false;
        int sg = 0;
#pragma unroll
        for (int sgi = 0; sgi + 1 < GEMM_MAX_SEG; ++sgi) sg += t >= seg_end_[sgi] ? 1 : 0;
        return ((pairbits >> sg) & 1u) != 0u;
    };
    // XCD-banded tile map: XCD x = b % 8 owns 8 consecutive row tiles (all column tiles), so a band's A panel is read from
    // HBM by one XCD and re-used from its L2 by the 4 column tiles
    const int b = blockIdx.x, xcd = b & 7, l = b >> 3;
    const int per = (p.tiles_m + 7) >> 3;
    const int tm = xcd * per + l / p.tiles_n, tn = l % p.tiles_n;
    if (l >= per * p.tiles_n || tm >= p.tiles_m) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef DAE_DW_PROBE
    const int nk = (DAE_DW_PROBE & 1) ? 1 : p.ktiles_total;                 // probe: one K tile only
#else
    const int nk = p.ktiles_total;
#endif
    const int nk0 = p.seg[0].ktiles;                                        // XBITS: K tiles whose A operand is built from the bit image
    const int row0_m = tm * DW_BM, row0_n = tn * BN;
    constexpr bool UPDATE = OPT != DW_GRAD_ONLY;
#ifdef DAE_DW_PROBE
    constexpr bool PREFETCH_W = (OPT == DAE_OPT_SGD) && !(DAE_DW_PROBE & 8);   // probe: no master-weight read
#else
    constexpr bool PREFETCH_W = (OPT == DAE_OPT_SGD);
#endif
    const int g = lane >> 5, c = lane & 31;
    float* __restrict__ Wp = e.W;
    f32x16 acc[DW_MB];
    // Epilogue work items: the 160 x 128 tile as 40 x 32 blocks of 4 rows x 4 columns, block `tid + 512 i` to thread tid (i < 3; the last
    // round is half full): ALL 8 waves run the optimizer, every global access is a 16-byte piece of a 512-byte row run.  (Before, the four
    // MFMA waves alone updated W straight from the accumulator layout -- 80 dword loads + 80 dword stores + 100 LDS writes per lane, one
    // wave per SIMD, the producers idle: the master-weight store alone cost 19 of the kernel's 51 us and the read 9, probe builds in
    // profiles/r03_experiments.md; a store of that form streams as fast as any other when nothing else limits it, tools/lds_stream_ubench.)
    constexpr int DW_EB = 3;
    // plain SGD: the master weights of this thread's blocks are requested BEFORE the K loop, by all 8 waves (in the producers the plain
    // loads are older than every LDS-DMA piece, so the counted vmcnt waits of the ring still hold)
    f32x4 wq[DW_EB][4];
    if constexpr (PREFETCH_W) {
#pragma unroll
        for (int i = 0; i < DW_EB; ++i) {
            const int blk = tid + PC_THREADS * i, rg = blk >> 5, c4 = blk & 31;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int grow = min(row0_m + rg * 4 + q, Mrows - 1);
                if (blk < DW_BM * 8) wq[i][q] = *reinterpret_cast<const f32x4*>(Wp + (int64_t)grow * e.ldw + row0_n + c4 * 4);
            }
        }
    }

    if (wave8 >= 4) {
        // ================= producer: per K tile 5 A pieces (DMA, or built from bits) + 4 B pieces =================
        const int wave = wave8 - 4;
        // XBITS: this wave builds tile rows [40 wave, +40): item 0 of a lane = (row 40 wave + (lane >> 1), 32-column half lane & 1),
        // item 1 (lanes 0..15) = (row 40 wave + 32 + (lane >> 1), half).  The lane's bit words of EVERY K tile of the segment are
        // loaded here, once, into registers that rotate by one per built stage (no run-time register index, and no ordinary
        // load inside the LDS-DMA loop -- hipcc would drain the DMA queue at its use)
        uint32_t bw0[DWB_MAXKT], bw1[DWB_MAXKT];
        const int half = lane & 1;
        const int lrow_a = wave * 40 + (lane >> 1), lrow_b = wave * 40 + 32 + (lane >> 1);
        if constexpr (XBITS) {
            const bool ok_a = row0_m + lrow_a < Mrows, ok_b = lane < 16 && row0_m + lrow_b < Mrows;
            const uint32_t* pa = xb.xtb + (int64_t)min(row0_m + lrow_a, Mrows - 1) * xb.ldxt;
            const uint32_t* pb = xb.xtb + (int64_t)min(row0_m + lrow_b, Mrows - 1) * xb.ldxt;
#pragma unroll
            for (int t = 0; t < DWB_MAXKT; ++t) {
                const int wi = min(2 * t + half, xb.nwords - 1);
                bw0[t] = pa[wi]; bw1[t] = pb[wi];
            }
#pragma unroll
            for (int t = 0; t < DWB_MAXKT; ++t) {
                const bool in = 2 * t + half < xb.nwords;
                bw0[t] = (in && ok_a) ? bw0[t] : 0u;
                bw1[t] = (in && ok_b) ? bw1[t] : 0u;
            }
        }
        uint32_t voA[5], voB[4];
        const char *gA = nullptr, *gB = nullptr, *gB2 = nullptr;
        int64_t a_adv = BKB;                                                 // bytes the A stream advances per K tile (TRA: 64 batch rows)
        int kt_dma = 0, seg_end = 0;
        auto seg_setup = [&](int kt) {
            int k;
            const int sg = seg_locate(p, kt, k, seg_end);
            const uint32_t lda = (uint32_t)p.seg[sg].lda_b, ldb = (uint32_t)p.seg[sg].ldb_b;
            if constexpr (PAIR) gB2 = p.bt2[sg] ? p.bt2[sg] + (int64_t)k * BKB : nullptr;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if constexpr (TRA) {       // chunk c of the linear [64 k][20 chunks of 8 features] image: batch row c / 20 of the K tile, features 8 (c % 20) ..
                    const int cch = (i * 4 + wave) * 64 + lane, kr = cch / 20, mc = cch % 20;
                    voA[i] = (uint32_t)kr * lda + (uint32_t)min(row0_m + mc * 8, Mrows - 8) * 2u;
                } else {
                const int row = (i * 4 + wave) * 8 + (lane >> 3);
                const int grow = min(row0_m + row, Mrows - 1);
                voA[i] = (uint32_t)grow * lda + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
                }
            }
            a_adv = TRA ? (int64_t)64 * lda : (int64_t)BKB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (i * 4 + wave) * 8 + (lane >> 3);
                voB[i] = (uint32_t)(row0_n + row) * ldb + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
            }
            gA = p.seg[sg].A + (int64_t)k * a_adv;
            gB = p.seg[sg].Bt + (int64_t)k * BKB;
        };
        seg_setup(0);
        // one (row, half) item of the A tile of the stage in `slot`: 32 consecutive k of tile row `lrow` from one bit word
        auto build_item = [&](char* slot, int lrow, uint32_t word, bool dense) {
            const uint32_t swz = (uint32_t)((lrow >> 1) & 7);
            if (dense) {                                   // arithmetic expansion: 8 bits -> 8 bf16 (0 / one) per 16-byte slot
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    i32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t b2 = (word >> (8 * j + 2 * q)) & 3u;
                        v[q] = (int)(((b2 & 1u) * xb.one) | ((b2 >> 1) * (xb.one << 16)));
                    }
                    *reinterpret_cast<i32x4*>(slot + lrow * BKB + (((uint32_t)(half * 4 + j) ^ swz) << 4)) = v;
                }
            } else {
                while (word) {                             // one 2-byte store per set bit (in-order LDS: lands after the zero fill)
                    const int bb = __builtin_ctz(word);
                    word &= word - 1;
                    const uint32_t k = (uint32_t)(half * 32 + bb);
                    *reinterpret_cast<bf16_t*>(slot + lrow * BKB + (((k >> 3) ^ swz) << 4) + (k & 7) * 2) = (bf16_t)xb.one;
                }
            }
        };
        auto build_a = [&](char* slot) {                   // consumes bw0[0] / bw1[0] and rotates the registers
            const uint32_t w0 = bw0[0], w1 = bw1[0];
#pragma unroll
            for (int t = 0; t + 1 < DWB_MAXKT; ++t) { bw0[t] = bw0[t + 1]; bw1[t] = bw1[t + 1]; }
            const bool dense = __builtin_amdgcn_ballot_w64(__builtin_popcount(w0) > 6 || __builtin_popcount(w1) > 6) != 0ull;
            if (!dense) {                                  // zero this wave's 40 rows (5 KiB): 5 x ds_write_b128 per lane
                const i32x4 z = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 5; ++i) *reinterpret_cast<i32x4*>(slot + wave * 40 * BKB + i * 1024 + lane * 16) = z;
            }
            build_item(slot, lrow_a, w0, dense);
            if (lane < 16) build_item(slot, lrow_b, w1, dense);
        };
        auto dma_stage = [&](char* slot) {                 // returns nothing; issues 9 (dense A) or 4 (built A) pieces
            const bool built = XBITS && kt_dma < nk0;
#if defined(DAE_DW_PROBE) && (DAE_DW_PROBE & 16)
            if (true) { ++kt_dma; return; }                // probe: no operand stream (the consumers multiply whatever the ring holds)
#endif
            if (!built) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                                     (__attribute__((address_space(3))) void*)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + voB[i]),
                                                 (__attribute__((address_space(3))) void*)(slot + DW_A_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
            if constexpr (PAIR) {
                if (gB2) {                                                  // the segment's second B tile (same rows, same leading dimension)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB2 + voB[i]),
                                                         (__attribute__((address_space(3))) void*)(slot + DW_A_BYTES + TILE_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
                }
            }
            if (built) build_a(slot);
            ++kt_dma;
            if (kt_dma == seg_end) { if (kt_dma < p.ktiles_total) seg_setup(kt_dma); }
            else { gA += a_adv; gB += BKB; if constexpr (PAIR) { if (gB2) gB2 += BKB; } }
        };
        auto ops = [&](int st) { return st >= nk ? 0 : ((XBITS && st < nk0) ? 4 : (pair_of(st) ? 13 : 9)); };   // LDS-DMA pieces of stage st (per wave)
#pragma unroll
        for (int st = 0; st < NSTG; ++st)
            if (st < nk) dma_stage(lds + st * STG);
        wait_vm_n(ops(1) + ops(2) + (PAIR ? 0 : ops(3)));                   // stage 0 landed (older plain loads return first)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // ... and every A tile built so far is written
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int i = 0; i < nk; ++i) {
            wait_vm_n(ops(i + 2) + (PAIR ? 0 : ops(i + 3)));                // stage i+1 landed; younger stages stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (i + NSTG < nk) dma_stage(lds + cur * STG);
            cur = cur + 1 == NSTG ? 0 : cur + 1;
        }
    } else {
        // ================= consumer =================
        const int wave = wave8;                                             // column block
#pragma unroll
        for (int m = 0; m < DW_MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        const int r = lane & 31;
        const int swz = (r >> 1) & 7;
        const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
        const uint32_t offa = r * BKB, offb = DW_A_BYTES + (wave * 32 + r) * BKB;
        uint32_t so[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
        i32x4 fa[2][DW_MB], fb[2], fb2[PAIR ? 2 : 1];
        // PAIR: every k step also reads a fragment of the stage's SECOND B tile (for an unpaired stage: the first tile's fragment once more, never
        // multiplied) -- the read count per step is then the same for both kinds of stage and the counted lgkmcnt waits stay compile-time
        constexpr uint32_t kOffB2 = DW_A_BYTES + TILE_BYTES;
        bool pair_cur = pair_of(0), pair_nxt = false;
        uint32_t ob2_cur = (pair_cur ? kOffB2 : (uint32_t)DW_A_BYTES) + (wave * 32 + r) * BKB, ob2_nxt = ob2_cur;
        (void)pair_nxt; (void)ob2_nxt;
        // TRA: per-lane base of the transposing reads -- k row 8 g + (i >> 2) of the k step, features 16 (q & 1) + 4 (i & 3) of the 32-row block (q = lane >> 4,
        // i = lane & 15); k step KK, half t and row block mb are immediate offsets (KK * 16 + 4 t rows of 320 B, mb * 64 B)
        const uint32_t ta = (uint32_t)((g * 8 + ((lane & 15) >> 2)) * DW_TR_PITCH + ((((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2));
#define DAE_DW_TR(S, KK, SLOTBASE, MB)                                                                                          \
    { const i32x2 lo__ = lds_read_tr16_b64<(KK) * 16 * DW_TR_PITCH + (MB) * 64>((SLOTBASE) + ta);                               \
      const i32x2 hi__ = lds_read_tr16_b64<((KK) * 16 + 4) * DW_TR_PITCH + (MB) * 64>((SLOTBASE) + ta);                         \
      fa[S][MB] = i32x4{lo__.x, lo__.y, hi__.x, hi__.y}; }
#define DAE_DW_READ(S, KK, SLOTBASE, OB2)                                              \
    fb[S] = lds_read_b128((SLOTBASE) + offb + so[KK]);                                 \
    if constexpr (PAIR) fb2[PAIR ? S : 0] = lds_read_b128((SLOTBASE) + (OB2) + so[KK]); \
    if constexpr (TRA) {                                                               \
        DAE_DW_TR(S, KK, SLOTBASE, 0) DAE_DW_TR(S, KK, SLOTBASE, 1) DAE_DW_TR(S, KK, SLOTBASE, 2)                               \
        DAE_DW_TR(S, KK, SLOTBASE, 3) DAE_DW_TR(S, KK, SLOTBASE, 4)                    \
    } else {                                                                           \
    fa[S][0] = lds_read_b128((SLOTBASE) + offa + so[KK]);                              \
    fa[S][1] = lds_read_b128_off4096((SLOTBASE) + offa + so[KK]);                      \
    fa[S][2] = lds_read_b128((SLOTBASE) + offa + 8192 + so[KK]);                       \
    fa[S][3] = lds_read_b128_off4096((SLOTBASE) + offa + 8192 + so[KK]);               \
    fa[S][4] = lds_read_b128((SLOTBASE) + offa + 16384 + so[KK]);                      \
    }
#define DAE_DW_MMA(S)                                                                  \
    Mma<bf16_t>::run(fa[S][0], fb[S], acc[0]);                                         \
    Mma<bf16_t>::run(fa[S][1], fb[S], acc[1]);                                         \
    Mma<bf16_t>::run(fa[S][2], fb[S], acc[2]);                                         \
    Mma<bf16_t>::run(fa[S][3], fb[S], acc[3]);                                         \
    Mma<bf16_t>::run(fa[S][4], fb[S], acc[4]);                                         \
    if constexpr (PAIR) {                                                              \
        if (pair_cur) {                                                                \
            Mma<bf16_t>::run(fa[S][0], fb2[PAIR ? S : 0], acc[0]);                     \
            Mma<bf16_t>::run(fa[S][1], fb2[PAIR ? S : 0], acc[1]);                     \
            Mma<bf16_t>::run(fa[S][2], fb2[PAIR ? S : 0], acc[2]);                     \
            Mma<bf16_t>::run(fa[S][3], fb2[PAIR ? S : 0], acc[3]);                     \
            Mma<bf16_t>::run(fa[S][4], fb2[PAIR ? S : 0], acc[4]);                     \
        }                                                                              \
    }
#define DAE_DW_WAIT_SET()                                                              \
    if constexpr (PAIR) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");             \
    else if constexpr (TRA) asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory");        \
    else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
#if defined(DAE_DW_PROBE) && (DAE_DW_PROBE & 32)           // probe: the consumers only take part in the barriers
#undef DAE_DW_READ
#undef DAE_DW_MMA
#define DAE_DW_READ(S, KK, SLOTBASE, OB2) fb[S] = i32x4{0, 0, 0, 0}; fa[S][0] = fa[S][1] = fa[S][2] = fa[S][3] = fa[S][4] = fb[S];
#define DAE_DW_MMA(S)
#elif defined(DAE_DW_PROBE) && (DAE_DW_PROBE & 64)          // probe: fragment reads but no MFMAs
#undef DAE_DW_MMA
#define DAE_DW_MMA(S) asm volatile("" :: "v"(fa[S][0]), "v"(fa[S][1]), "v"(fa[S][2]), "v"(fa[S][3]), "v"(fa[S][4]), "v"(fb[S]));
#elif defined(DAE_DW_PROBE) && (DAE_DW_PROBE & 128)         // probe: MFMAs on whatever the registers hold, no fragment reads
#undef DAE_DW_READ
#define DAE_DW_READ(S, KK, SLOTBASE, OB2) asm volatile("" : "+v"(fa[S][0]), "+v"(fa[S][1]), "+v"(fa[S][2]), "+v"(fa[S][3]), "+v"(fa[S][4]), "+v"(fb[S]));
#endif
        __builtin_amdgcn_s_barrier();                                       // stage 0 landed (producers waited for it)
        asm volatile("" ::: "memory");
        DAE_DW_READ(0, 0, lbase, ob2_cur)
        __builtin_amdgcn_sched_barrier(0);
        int cur = 0;
        for (int i = 0; i < nk; ++i) {
            const int nxt = cur + 1 == NSTG ? 0 : cur + 1;
            const uint32_t sb = lbase + cur * STG, nb = lbase + nxt * STG;
            if constexpr (PAIR) {                                           // kind of the NEXT stage (its first k step is read behind the barrier below)
                pair_nxt = (i + 1 < nk) && pair_of(i + 1);
                ob2_nxt = (pair_nxt ? kOffB2 : (uint32_t)DW_A_BYTES) + (wave * 32 + r) * BKB;
            }
            DAE_DW_READ(1, 1, sb, ob2_cur)                                   // k step 1 -> set 1
            DAE_DW_WAIT_SET()                                                // k step 0 (set 0) landed
            __builtin_amdgcn_sched_barrier(0);
            DAE_DW_MMA(0)
            __builtin_amdgcn_sched_barrier(0);
            DAE_DW_READ(0, 2, sb, ob2_cur)
            DAE_DW_WAIT_SET()
            __builtin_amdgcn_sched_barrier(0);
            DAE_DW_MMA(1)
            __builtin_amdgcn_sched_barrier(0);
            DAE_DW_READ(1, 3, sb, ob2_cur)
            DAE_DW_WAIT_SET()
            __builtin_amdgcn_sched_barrier(0);
            DAE_DW_MMA(0)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // every LDS read of tile i is done
            __builtin_amdgcn_s_barrier();                                   // slot free for the producers; stage i+1 landed
            asm volatile("" ::: "memory");
            DAE_DW_READ(0, 0, nb, ob2_nxt)                                   // stale (never consumed) after the last tile
            __builtin_amdgcn_sched_barrier(0);
            DAE_DW_MMA(1)                                                    // (k step 3 of tile i: pair_cur is still tile i's)
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
            if constexpr (PAIR) { pair_cur = pair_nxt; ob2_cur = ob2_nxt; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef DAE_DW_READ
#undef DAE_DW_TR
#undef DAE_DW_MMA
#undef DAE_DW_WAIT_SET
    }
    __builtin_amdgcn_s_barrier();                                           // B1: every wave is out of the K loop; the ring is dead
    asm volatile("" ::: "memory");

    // ---- epilogue: the gradient tile is parked in LDS (fp32 [160][128], conflict-free from the accumulator layout), then all 8 waves
    //      run the optimizer block-wise: W, optimizer slots and the optional gradient image as 16-byte pieces, W_lo as 8-byte pieces;
    //      the transposed shadow is staged in a second LDS tile and leaves in 16-byte pieces ----
    float* Gt = reinterpret_cast<float*>(lds);                              // [160][128] fp32 (80 KiB)
    char* R1 = lds + DW_BM * 128 * 4;                                       // Wt_lo tile  [128][DW_P1] (42 KiB)
    static_assert(DW_BM * 128 * 4 + 128 * DW_P1 <= DW_RING && DW_BM * 128 * 4 + 128 * DW_P1 <= DW_RING2, "the epilogue tiles must fit the dead ring");
    static_assert(DW_RING2 <= 160 * 1024, "the paired ring must fit the CU's LDS");
    if (wave8 < 4) {
        const int lcol = wave8 * 32 + c;
#pragma unroll
        for (int m = 0; m < DW_MB; ++m)
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) Gt[(m * 32 + (r2 & 3) + 8 * (r2 >> 2) + 4 * g) * 128 + lcol] = acc[m][r2];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                           // B2: the gradient tile is complete
    asm volatile("" ::: "memory");
    float pvk[X3 ? DW_EB : 1][4][4];                                        // X3: the updated weights stay in registers for the lo round below
    {
        float* __restrict__ gradp = e.grad;
        float* __restrict__ s1p = e.s1;
        float* __restrict__ s2p = e.s2;
        bf16_t* __restrict__ Wlo = reinterpret_cast<bf16_t*>(UPDATE ? e.W_lo : e.grad_lo);
        bf16_t* __restrict__ Wlo2 = reinterpret_cast<bf16_t*>(e.W_lo2);
        const float lr = e.lr, mom = e.mom, gscale = e.gscale, gin = e.gin == 0.f ? 1.f : e.gin;
#pragma unroll
        for (int i = 0; i < DW_EB; ++i) {
            const int blk = tid + PC_THREADS * i, rg = blk >> 5, c4 = blk & 31;
            if (blk >= DW_BM * 8) break;
            float (&pv)[4][4] = pvk[X3 ? i : 0];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lrow = rg * 4 + q;
                const bool ok = row0_m + lrow < Mrows;
                const int64_t k = (int64_t)min(row0_m + lrow, Mrows - 1) * e.ldw + row0_n + c4 * 4;
                const f32x4 gr = *reinterpret_cast<const f32x4*>(Gt + lrow * 128 + c4 * 4) * gin;     // un-scale the 16-bit delta images' power of two
                if constexpr (!UPDATE) {
                    if (ok && gradp) *reinterpret_cast<f32x4*>(gradp + k) = gr;
                    if (ok && Wlo) {
                        uint2 lo;
                        lo.x = f2bf_pack_hw(gr[0], gr[1]); lo.y = f2bf_pack_hw(gr[2], gr[3]);
                        *reinterpret_cast<uint2*>(Wlo + k) = lo;
                    }
                    continue;
                }
                f32x4 p0, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, pn;
                if constexpr (PREFETCH_W) p0 = wq[i][q];
                else p0 = *reinterpret_cast<const f32x4*>(Wp + k);
                if constexpr (OPT == DAE_OPT_ADAGRAD || OPT == DAE_OPT_MOMENTUM || OPT == DAE_OPT_ADAM) a1 = *reinterpret_cast<const f32x4*>(s1p + k);
                if constexpr (OPT == DAE_OPT_ADAM) a2 = *reinterpret_cast<const f32x4*>(s2p + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gg = gr[j] * gscale;
                    if constexpr (OPT == DAE_OPT_SGD) pn[j] = p0[j] - lr * gg;
                    else if constexpr (OPT == DAE_OPT_ADAGRAD) { const float a = a1[j] + gg * gg; a1[j] = a; pn[j] = p0[j] - lr * gg * rsqrtf(a); }
                    else if constexpr (OPT == DAE_OPT_MOMENTUM) { const float a = mom * a1[j] + gg; a1[j] = a; pn[j] = p0[j] - lr * a; }
                    else {
                        const float mm = 0.9f * a1[j] + 0.1f * gg;
                        const float vv = 0.999f * a2[j] + 0.001f * gg * gg;
                        a1[j] = mm; a2[j] = vv;
                        pn[j] = p0[j] - lr * mm / (sqrtf(vv) + 1e-8f);
                    }
                    pv[q][j] = pn[j];
                }
                if (ok) {
#ifdef DAE_DW_PROBE
                    if (!(DAE_DW_PROBE & 2))                                 // probe: no master-weight store
#endif
                    *reinterpret_cast<f32x4*>(Wp + k) = pn;
                    if (gradp) *reinterpret_cast<f32x4*>(gradp + k) = gr;
                    if constexpr (OPT == DAE_OPT_ADAGRAD || OPT == DAE_OPT_MOMENTUM || OPT == DAE_OPT_ADAM) *reinterpret_cast<f32x4*>(s1p + k) = a1;
                    if constexpr (OPT == DAE_OPT_ADAM) *reinterpret_cast<f32x4*>(s2p + k) = a2;
#ifdef DAE_DW_PROBE
                    if (!(DAE_DW_PROBE & 4))                                 // probe: no shadow stores
#endif
                    {
                        uint2 lo;
                        lo.x = f2bf_pack_hw(pn[0], pn[1]); lo.y = f2bf_pack_hw(pn[2], pn[3]);
                        *reinterpret_cast<uint2*>(Wlo + k) = lo;
                        if constexpr (X3) {
                            if (Wlo2) {                                      // NULL: nobody reads the lo image of the row-major shadow
                                uint2 l2;
                                l2.x = bf_residual_pack_hw(pn[0], pn[1]); l2.y = bf_residual_pack_hw(pn[2], pn[3]);
                                *reinterpret_cast<uint2*>(Wlo2 + k) = l2;
                            }
                        }
                    }
                }
            }
            if constexpr (UPDATE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {                               // 4 features of column c4 * 4 + j: one 8-byte piece of the transposed tile
                    uint2 v;
                    v.x = f2bf_pack_hw(pv[0][j], pv[1][j]);
                    v.y = f2bf_pack_hw(pv[2][j], pv[3][j]);
                    *reinterpret_cast<uint2*>(R1 + (c4 * 4 + j) * DW_P1 + rg * 8) = v;
                }
            }
        }
    }
    if constexpr (!UPDATE) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                           // B3: the transposed tile is complete
    asm volatile("" ::: "memory");
#ifdef DAE_DW_PROBE
    if (DAE_DW_PROBE & 4) return;                                           // probe: no shadow stores
#endif
    {
        bf16_t* Wtlo = reinterpret_cast<bf16_t*>(e.Wt_lo);
        for (int ch = tid; ch < 128 * 20; ch += PC_THREADS) {                // Wt_lo: 128 rows x 20 chunks of 8 features
            const int row = ch / 20, c16 = ch % 20;
            if (row0_m + c16 * 8 < Mrows)
                *reinterpret_cast<i32x4*>(Wtlo + (int64_t)(row0_n + row) * e.ldwt + row0_m + c16 * 8) =
                    *reinterpret_cast<const i32x4*>(R1 + row * DW_P1 + c16 * 16);
        }
    }
    if constexpr (X3) {                                                     // second round through the same staging tile: Wt_lo2 = bf16(W - bf16(W))^T
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // every piece of the hi tile has been read
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < DW_EB; ++i) {
            const int blk = tid + PC_THREADS * i, rg = blk >> 5, c4 = blk & 31;
            if (blk >= DW_BM * 8) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 v;
                v.x = bf_residual_pack_hw(pvk[i][0][j], pvk[i][1][j]);
                v.y = bf_residual_pack_hw(pvk[i][2][j], pvk[i][3][j]);
                *reinterpret_cast<uint2*>(R1 + (c4 * 4 + j) * DW_P1 + rg * 8) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16_t* Wtlo2 = reinterpret_cast<bf16_t*>(e.Wt_lo2);
        for (int ch = tid; ch < 128 * 20; ch += PC_THREADS) {
            const int row = ch / 20, c16 = ch % 20;
            if (row0_m + c16 * 8 < Mrows)
                *reinterpret_cast<i32x4*>(Wtlo2 + (int64_t)(row0_n + row) * e.ldwt + row0_m + c16 * 8) =
                    *reinterpret_cast<const i32x4*>(R1 + row * DW_P1 + c16 * 16);
        }
    }
}

// Instrumented twin of gemm_nt_f32out (dae_gemm_trace): same code with shader-clock stamps; o[5] = K loop, o[6] = epilogue,
// o[7] = s_memtime at kernel entry (block start skew).
template <typename T, int NST>
__global__ __launch_bounds__(GEMM_THREADS, wg_per_cu_for(NST)) void gemm_nt_trace(GemmParams p, float* __restrict__ C, int64_t ldc,
                                                                                  int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int tm, tn, split, kt0, kt1;
    if (!block_to_tile(p, tm, tn, split, kt0, kt1)) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 acc[2][2];
    gemm_mainloop<T, NST, true>(p, tm, tn, kt0, kt1, lds, acc);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        unsigned long long* o = p.trace + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[5] = t1 - t0; o[6] = t2 - t1; o[7] = t0;
    }
}

// ------------------------------------------------------------------------------------------------
// decode kernel: GEMM + bias + activation + reconstruction loss + d cost/d z2 (+ bias-gradient partials)
// autoencoder.py:411, triplet_loss_utils.py:262-277
//
// Epilogue data flow (bf16, "STAGED"): the clean-input tile x[128x128] is prefetched into registers with
// coalesced 16-byte loads BEFORE the K loop (its HBM latency hides under the MFMAs), parked in LDS after the
// loop and read per accumulator element from there; delta2 is written back into the same LDS cell, delta2^T
// into a second LDS tile, and both leave the CU as coalesced 16-byte row segments.  Row sums (loss) and
// column sums (db_v) are accumulated with LDS float atomics whose addresses are private to one wave, so the
// addition order -- and the result -- is deterministic.  fp32 (parity mode) keeps direct global accesses.
// LOSS / ACT are compile-time so the hot specialisation (cross_entropy + sigmoid) carries no dead code.
// ------------------------------------------------------------------------------------------------
// Tile geometry of the decode kernel: 128 rows x BN_T columns.  BN_T = 64 (bf16) gives 7 x 158 = 1106 tiles of 48 KB LDS
// and <= 168 VGPRs, i.e. THREE workgroups per CU and 768 resident slots: the 128 x 128 form had 553 tiles on 512 slots
// (2 per CU), so 41 stragglers doubled the kernel time (rocprofv3: wave lifetime 15.6 us, kernel 32 us).
template <int BN_T> struct DecGeo {
    static constexpr int NTB = BN_T / 64;                  // 32-column MFMA blocks per wave along N (waves are 2 x 2)
    static constexpr int WCOLS = BN_T / 2;                 // columns per wave
    static constexpr int P0 = BN_T * 2 + 16;               // staged row pitch of the [128][BN_T] delta2 tile (bf16 + 16 B pad)
    static constexpr int P1 = 128 * 2 + 16;                // staged row pitch of the [BN_T][128] delta2^T tile
    static constexpr int R0_BYTES = 128 * P0;
    static constexpr int R1_BYTES = BN_T * P1;
    static constexpr int AUX_OFF = R0_BYTES + R1_BYTES;
    static constexpr int AUX_FLOATS = 13 * 128 + 128 * (BN_T / 32);   // 13 x 128 floats + the 128 x (BN_T/32) words of the x bit tile
    static constexpr int EPI_BYTES = AUX_OFF + AUX_FLOATS * 4;
    static constexpr int STAGE = TILE_BYTES + BN_T * BKB;  // K-loop stage: A tile 16 KB + B tile
    static constexpr int LOOP_BYTES = BN_T == 128 ? lds_bytes_for(2) : 2 * STAGE;
    static constexpr int LDS_BYTES = LOOP_BYTES > EPI_BYTES ? LOOP_BYTES : EPI_BYTES;
    static constexpr int WG_PER_CU = BN_T == 128 ? 2 : 3;
};

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

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

template <int ACT> __device__ __forceinline__ float act_fwd(float z) {
    if constexpr (ACT == DAE_ACT_SIGMOID) {
        const float en = __builtin_amdgcn_exp2f(-fabsf(z) * kLog2e);      // exp(-|z|) in (0,1]
        const float r = __builtin_amdgcn_rcpf(1.0f + en);
        return z >= 0.f ? r : en * r;
    } else if constexpr (ACT == DAE_ACT_TANH) {
        const float e2 = __builtin_amdgcn_exp2f(-2.0f * fabsf(z) * kLog2e);   // exp(-2|z|)
        const float t = (1.0f - e2) * __builtin_amdgcn_rcpf(1.0f + e2);
        return z >= 0.f ? t : -t;
    } else {
        return z;
    }
}
template <int ACT> __device__ __forceinline__ float act_bwd(float a) {
    if constexpr (ACT == DAE_ACT_SIGMOID) return a * (1.0f - a);
    else if constexpr (ACT == DAE_ACT_TANH) return 1.0f - a * a;
    else return 1.0f;
}

// K loop of the 128 x 64 decode tile: two LDS stages filled by global_load_lds (A 4 + B 2 pieces of 1 KiB per wave), two
// raw barriers per K tile.  With three workgroups per CU the DMA latency and the barriers of one workgroup hide behind the
// MFMAs / epilogue VALU of the other two, so the loop itself stays simple; K = Hp is only 8 tiles deep.
// Waves 2 x 2: wave (wm, wn) owns rows [64 wm, +64) x columns [32 wn, +32) = 2 MFMA 32x32 accumulators.
template <typename T>
__device__ __forceinline__ void mainloop_n64(const GemmParams& p, int tm, int tn, char* lds, f32x16 (&acc)[2][1]) {
    constexpr int BN_T = 64, STAGE = DecGeo<64>::STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int row0_m = tm * BM, row0_n = tn * BN_T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const int nk = p.ktiles_total;                     // all K segments back to back, no split-K
    if (nk <= 0) return;
    uint32_t voA[4], voB[2];
    const char *gA = nullptr, *gB = nullptr;
    int kt_dma = 0, seg_end = 0;
    auto seg_setup = [&](int kt) {
        int k;
        const int sg = seg_locate(p, kt, k, seg_end);
        const uint32_t lda = (uint32_t)p.seg[sg].lda_b, ldb = (uint32_t)p.seg[sg].ldb_b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 4 + wave) * 8 + (lane >> 3);
            voA[i] = (uint32_t)(row0_m + row) * lda + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 4 + wave) * 8 + (lane >> 3);
            voB[i] = (uint32_t)(row0_n + row) * ldb + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        gA = p.seg[sg].A + (int64_t)k * BKB;
        gB = p.seg[sg].Bt + (int64_t)k * BKB;
    };
    seg_setup(0);
    auto dma_stage = [&](char* slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                             (__attribute__((address_space(3))) void*)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
        ++kt_dma;
        if (kt_dma == seg_end) { if (kt_dma < nk) seg_setup(kt_dma); }
        else { gA += BKB; gB += BKB; }
    };
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const uint32_t offa = (wm * 64 + r) * BKB, offb = TILE_BYTES + (wn * 32 + r) * BKB;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
    dma_stage(lds);
    for (int i = 0; i < nk; ++i) {
        if (i + 1 < nk) { dma_stage(lds + ((i + 1) & 1) * STAGE); wait_vm<6>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                  // stage i landed for every wave
        asm volatile("" ::: "memory");
        const uint32_t sb = lbase + (i & 1) * STAGE;
        i32x4 fa[4][2], fb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[kk][0] = lds_read_b128(sb + offa + so[kk]);
            fa[kk][1] = lds_read_b128_off4096(sb + offa + so[kk]);
            fb[kk] = lds_read_b128(sb + offb + so[kk]);
        }
#define DAE_N64_GROUP(KK, CNT)                                   \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<T>::run(fa[KK][0], fb[KK], acc[0][0]);                   \
    Mma<T>::run(fa[KK][1], fb[KK], acc[1][0]);
        DAE_N64_GROUP(0, 9)
        DAE_N64_GROUP(1, 6)
        DAE_N64_GROUP(2, 3)
        DAE_N64_GROUP(3, 0)
#undef DAE_N64_GROUP
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // every wave has read slot i & 1: iteration i+1 may refill it
        asm volatile("" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------------
// Gram matrix D = h h^T of the 16-bit modes (split operands [hi | hi | lo] . [hi | lo | hi]^T, K = 3 Hp) on 64 x 64 tiles, ONE slab.
// The 128 x 128 / split-K-4 form (gemm_nt_pc<GRAM>) is pure latency at this size -- 196 workgroups of 6 K tiles each, 4 slabs of 3.2 MB written
// and read back by the miner's prologue (12.5 MB + 10.6 MB per step, 13 us).  Here a workgroup owns a 64 x 64 tile of D over the WHOLE K (24 K
// tiles at H = 500): 196 workgroups = one per CU, four waves of one 32 x 32 accumulator each, an 8-stage LDS ring (16 KiB per stage) filled by
// LDS-DMA so that seven stages are in flight while one is multiplied -- the K loop never waits for memory after the first stage -- and D leaves
// once, as the sum the miner wants.  Tile map: XCD x = b % 8 takes a contiguous run of tiles (same row tile = same A panel in its L2).
// ------------------------------------------------------------------------------------------------
constexpr int G64_NST = 8;
constexpr int G64_TILE = 64 * BKB;                 // 8 KiB per operand per stage
constexpr int G64_STAGE = 2 * G64_TILE;
constexpr int G64_LDS = G64_NST * G64_STAGE;       // 128 KiB: one workgroup per CU
__device__ __forceinline__ void wait_vm_4n(int n) {   // vmcnt(4 n): n younger stages of 4 LDS-DMA pieces each may stay in flight
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<4>(); break;
        case 2: wait_vm<8>(); break;
        case 3: wait_vm<12>(); break;
        case 4: wait_vm<16>(); break;
        case 5: wait_vm<20>(); break;
        case 6: wait_vm<24>(); break;
        default: wait_vm<28>(); break;
    }
}
__global__ __launch_bounds__(GEMM_THREADS, 1) void gram64_kernel(const char* __restrict__ A, int64_t lda_b, const char* __restrict__ Bt, int64_t ldb_b,
                                                                 int nk, int tiles, float* __restrict__ D, int64_t ldd) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int b = blockIdx.x, per = (tiles * tiles + 7) >> 3;
    const int t = (b & 7) * per + (b >> 3);
    if ((b >> 3) >= per || t >= tiles * tiles) return;
    const int tm = t / tiles, tn = t % tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // LDS-DMA: this wave's pieces {wave, wave + 4} of each operand tile (8 rows of 128 B per 1-KiB piece), swizzled source slot per lane
    uint32_t voA[2], voB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const uint32_t ss = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        voA[i] = (uint32_t)(tm * 64 + row) * (uint32_t)lda_b + ss;
        voB[i] = (uint32_t)(tn * 64 + row) * (uint32_t)ldb_b + ss;
    }
    const char *gA = A, *gB = Bt;
    auto dma_stage = [&](char* slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                             (__attribute__((address_space(3))) void*)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + G64_TILE + (i * 4 + wave) * 1024), 16, 0, 0);
        }
        gA += BKB; gB += BKB;
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const uint32_t offa = (wm * 32 + r) * BKB, offb = G64_TILE + (wn * 32 + r) * BKB;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
#pragma unroll
    for (int st = 0; st < G64_NST; ++st)
        if (st < nk) dma_stage(lds + st * G64_STAGE);
    // ONE barrier per K tile: passing the barrier of iteration i proves that every wave has finished iteration i - 1 (its fragment reads included), so
    // the slot of stage i - 1 is refilled right behind it (stage i - 1 + NST) -- issued between this tile's fragment reads and its MFMAs, where the
    // ~50-cycle issue cost of each LDS-DMA piece hides the LDS latency of the reads
    int cur = 0, prev = G64_NST - 1;
    for (int i = 0; i < nk; ++i) {
        const int issued = i == 0 ? min(nk, G64_NST) : min(nk, i - 1 + G64_NST);      // stages requested so far
        wait_vm_4n(issued - (i + 1));                      // everything up to stage i has landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                      // ... and every other wave's; slot `prev` is free
        asm volatile("" ::: "memory");
        const uint32_t sb = lbase + cur * G64_STAGE;
        i32x4 fa[4], fb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[kk] = lds_read_b128(sb + offa + so[kk]);
            fb[kk] = lds_read_b128(sb + offb + so[kk]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i >= 1 && i - 1 + G64_NST < nk) dma_stage(lds + prev * G64_STAGE);
        __builtin_amdgcn_sched_barrier(0);
#define DAE_G64_STEP(KK, CNT)                                    \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<bf16_t>::run(fa[KK], fb[KK], acc);
        DAE_G64_STEP(0, 6)
        DAE_G64_STEP(1, 4)
        DAE_G64_STEP(2, 2)
        DAE_G64_STEP(3, 0)
#undef DAE_G64_STEP
        __builtin_amdgcn_sched_barrier(0);
        prev = cur;
        cur = cur + 1 == G64_NST ? 0 : cur + 1;
    }
    float* Dt = D + (int64_t)(tm * 64 + wm * 32) * ldd + tn * 64 + wn * 32 + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) Dt[(int64_t)((q & 3) + 8 * (q >> 2) + 4 * g) * ldd] = acc[q];
}

// The same tile with the THREE products of a K tile in ONE stage (round 6): a stage holds the hi and the lo image of both row panels for 64 columns of h
// (4 x 8 KiB), a wave reads its 16 fragments once and multiplies hi.hi, hi.lo, lo.hi from registers -- 8 stages of 8 LDS-DMA pieces / 16 fragment reads /
// one barrier per wave instead of 24 stages of 4 / 8 / one for the same 96 MFMAs, on three independent accumulators (the K-concatenated walk chains all
// 96 on one).  Same products; the three partial sums are added at the end.  4-stage ring of 32 KiB.
constexpr int G64F_NST = 4;
constexpr int G64F_STAGE = 4 * G64_TILE;
constexpr int G64F_LDS = G64F_NST * G64F_STAGE;
__device__ __forceinline__ void wait_vm_8n(int n) {   // vmcnt(8 n): n younger stages of 8 LDS-DMA pieces each may stay in flight
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<8>(); break;
        case 2: wait_vm<16>(); break;
        default: wait_vm<24>(); break;
    }
}
__global__ __launch_bounds__(GEMM_THREADS, 1) void gram64f_kernel(const char* __restrict__ Hhi, const char* __restrict__ Hlo, int64_t ld_b, int nk, int tiles,
                                                                  float* __restrict__ D, int64_t ldd) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int b = blockIdx.x, per = (tiles * tiles + 7) >> 3;
    const int t = (b & 7) * per + (b >> 3);
    if ((b >> 3) >= per || t >= tiles * tiles) return;
    const int tm = t / tiles, tn = t % tiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // LDS-DMA: this wave's pieces {wave, wave + 4} of each of the four operand tiles (8 rows of 128 B per 1-KiB piece), swizzled source slot per lane
    uint32_t voA[2], voB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const uint32_t ss = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        voA[i] = (uint32_t)(tm * 64 + row) * (uint32_t)ld_b + ss;
        voB[i] = (uint32_t)(tn * 64 + row) * (uint32_t)ld_b + ss;
    }
    const char *gH = Hhi, *gL = Hlo;
    auto dma_stage = [&](char* slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = (i * 4 + wave) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gH + voA[i]), (__attribute__((address_space(3))) void*)(slot + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gL + voA[i]), (__attribute__((address_space(3))) void*)(slot + G64_TILE + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gH + voB[i]), (__attribute__((address_space(3))) void*)(slot + 2 * G64_TILE + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gL + voB[i]), (__attribute__((address_space(3))) void*)(slot + 3 * G64_TILE + piece), 16, 0, 0);
        }
        gH += BKB; gL += BKB;
    };
    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const uint32_t offa = (wm * 32 + r) * BKB, offb = 2 * G64_TILE + (wn * 32 + r) * BKB;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
#pragma unroll
    for (int st = 0; st < G64F_NST; ++st)
        if (st < nk) dma_stage(lds + st * G64F_STAGE);
    int cur = 0, prev = G64F_NST - 1;
    for (int i = 0; i < nk; ++i) {
        const int issued = i == 0 ? min(nk, G64F_NST) : min(nk, i - 1 + G64F_NST);     // stages requested so far
        wait_vm_8n(issued - (i + 1));                      // everything up to stage i has landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                      // ... and every other wave's; slot `prev` is free
        asm volatile("" ::: "memory");
        const uint32_t sb = lbase + cur * G64F_STAGE;
        i32x4 fah[4], fal[4], fbh[4], fbl[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fah[kk] = lds_read_b128(sb + offa + so[kk]);
            fbh[kk] = lds_read_b128(sb + offb + so[kk]);
            fbl[kk] = lds_read_b128(sb + offb + G64_TILE + so[kk]);
            fal[kk] = lds_read_b128(sb + offa + G64_TILE + so[kk]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i >= 1 && i - 1 + G64F_NST < nk) dma_stage(lds + prev * G64F_STAGE);      // (its issue cost hides the LDS latency of the reads above)
        __builtin_amdgcn_sched_barrier(0);
#define DAE_G64F_STEP(KK, CNT)                                   \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<bf16_t>::run(fah[KK], fbh[KK], acc0);                    \
    Mma<bf16_t>::run(fah[KK], fbl[KK], acc1);                    \
    Mma<bf16_t>::run(fal[KK], fbh[KK], acc2);
        DAE_G64F_STEP(0, 12)
        DAE_G64F_STEP(1, 8)
        DAE_G64F_STEP(2, 4)
        DAE_G64F_STEP(3, 0)
#undef DAE_G64F_STEP
        __builtin_amdgcn_sched_barrier(0);
        prev = cur;
        cur = cur + 1 == G64F_NST ? 0 : cur + 1;
    }
    float* Dt = D + (int64_t)(tm * 64 + wm * 32) * ldd + tn * 64 + wn * 32 + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) Dt[(int64_t)((q & 3) + 8 * (q >> 2) + 4 * g) * ldd] = (acc0[q] + acc1[q]) + acc2[q];
}

static int g_gram_fused = 1;        // dae_set_glds(-19) off / (-20) on: the Gram's three products per K tile in one stage (gram64f_kernel) instead of the K-concatenated walk
int launch_gram64(const void* hcat_a, const void* hcat_b, int Bp, int Hp, float* D, hipStream_t st) {
    if (g_gram_fused && hcat_a && hcat_b && D && Bp % 64 == 0 && Hp % 64 == 0 && (uint64_t)Bp * (uint64_t)(3 * Hp * 2) < (1ull << 32)) {
        // hcat_a = [hi | hi | lo], hcat_b = [hi | lo | hi] (leading dimension 3 Hp): the hi image is block 0 of either, the lo image block 2 of hcat_a
        static int rcf = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gram64f_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G64F_LDS);
        DAE_CHECK_ARG(rcf == 0, "gram64: hipFuncSetAttribute failed");
        const int tiles = Bp / 64, per = (tiles * tiles + 7) / 8;
        DAE_LAUNCH(gram64f_kernel, dim3(8 * per), dim3(GEMM_THREADS), G64F_LDS, st, (const char*)hcat_a, (const char*)hcat_a + (size_t)2 * Hp * 2,
                           (int64_t)3 * Hp * 2, Hp * 2 / BKB, tiles, D, (int64_t)Bp);
        DAE_CHECK_LAUNCH();
        return 0;
    }
    DAE_CHECK_ARG(hcat_a && hcat_b && D && Bp % 64 == 0 && Hp % 64 == 0, "gram64: bad arguments");
    DAE_CHECK_ARG((uint64_t)Bp * (uint64_t)(3 * Hp * 2) < (1ull << 32), "gram64: operand panel beyond 4 GiB");
    static int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gram64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G64_LDS);
    DAE_CHECK_ARG(rc == 0, "gram64: hipFuncSetAttribute failed");
    const int tiles = Bp / 64, per = (tiles * tiles + 7) / 8;
    DAE_LAUNCH(gram64_kernel, dim3(8 * per), dim3(GEMM_THREADS), G64_LDS, st, (const char*)hcat_a, (int64_t)3 * Hp * 2, (const char*)hcat_b,
                       (int64_t)3 * Hp * 2, 3 * Hp * 2 / BKB, tiles, D, (int64_t)Bp);
    DAE_CHECK_LAUNCH();
    return 0;
}

// Paired form of the same loop for the split modes' two W terms, z2 = h.W_hi^T + h.W_lo^T: both K segments multiply the SAME A tile (h), so a stage
// holds ONE A tile and the B tiles of BOTH segments (16 + 8 + 8 KiB).  Per K tile pair a wave then issues 8 LDS-DMA pieces and 16 fragment reads for its
// 16 MFMAs instead of 12 and 24 -- the decode's K loop is bound by exactly those two (LDS bandwidth and LDS-DMA issue) -- at one barrier pair
// instead of two.  64 KiB of LDS per workgroup: two workgroups per CU instead of three.  The accumulation order differs from the unpaired walk
// ((hi, lo) interleaved per K tile instead of all hi then all lo): same products, fp32 sums in another order.
template <typename T>
__device__ __forceinline__ void mainloop_n64_pair(const GemmParams& p, int tm, int tn, char* lds, f32x16 (&acc)[2][1]) {
    constexpr int BT = 64 * BKB;                       // one B tile: 8 KiB
    constexpr int STAGE = TILE_BYTES + 2 * BT;         // 32 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int row0_m = tm * BM, row0_n = tn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const int nk = p.seg[0].ktiles;                    // both segments: the same K extent (checked by the launcher)
    if (nk <= 0) return;
    uint32_t voA[4], voB[2];
    const uint32_t lda = (uint32_t)p.seg[0].lda_b, ldb = (uint32_t)p.seg[0].ldb_b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        voA[i] = (uint32_t)(row0_m + row) * lda + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        voB[i] = (uint32_t)(row0_n + row) * ldb + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const char *gA = p.seg[0].A, *gB0 = p.seg[0].Bt, *gB1 = p.seg[1].Bt;
    auto dma_stage = [&](char* slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                             (__attribute__((address_space(3))) void*)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB0 + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB1 + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + BT + (i * 4 + wave) * 1024), 16, 0, 0);
        }
        gA += BKB; gB0 += BKB; gB1 += BKB;
    };
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const uint32_t offa = (wm * 64 + r) * BKB, offb = TILE_BYTES + (wn * 32 + r) * BKB;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
    dma_stage(lds);
    for (int i = 0; i < nk; ++i) {
        if (i + 1 < nk) { dma_stage(lds + ((i + 1) & 1) * STAGE); wait_vm<8>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                  // stage i landed for every wave
        asm volatile("" ::: "memory");
        const uint32_t sb = lbase + (i & 1) * STAGE;
        i32x4 fa[4][2], fb0[4], fb1[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[kk][0] = lds_read_b128(sb + offa + so[kk]);
            fa[kk][1] = lds_read_b128_off4096(sb + offa + so[kk]);
            fb0[kk] = lds_read_b128(sb + offb + so[kk]);
            fb1[kk] = lds_read_b128(sb + offb + BT + so[kk]);
        }
#define DAE_N64P_GROUP(KK, CNT)                                  \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<T>::run(fa[KK][0], fb0[KK], acc[0][0]);                  \
    Mma<T>::run(fa[KK][1], fb0[KK], acc[1][0]);                  \
    Mma<T>::run(fa[KK][0], fb1[KK], acc[0][0]);                  \
    Mma<T>::run(fa[KK][1], fb1[KK], acc[1][0]);
        DAE_N64P_GROUP(0, 12)
        DAE_N64P_GROUP(1, 8)
        DAE_N64P_GROUP(2, 4)
        DAE_N64P_GROUP(3, 0)
#undef DAE_N64P_GROUP
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // every wave has read slot i & 1: iteration i+1 may refill it
        asm volatile("" ::: "memory");
    }
}

// Three-term form for the split modes that keep h AND W as hi + lo (f16x2h: z2 = h_hi.W_hi^T + h_hi.W_lo^T + h_lo.W_hi^T).  The plain walk streams three K
// segments -- 3 stages per K tile, every h_hi and every W_hi tile fetched and read twice.  Here the stages alternate (h_hi, W_hi)[k] -> (h_lo, W_lo)[k]
// and the hi stage's fragments STAY IN REGISTERS (48 VGPRs: the kernel has the room at three workgroups per CU) over the lo stage, which multiplies
// (h_hi, W_lo) and (h_lo, W_hi): 2 stages per K tile -- 12 LDS-DMA pieces, 24 fragment reads, 4 barriers per wave -- for the same 24 MFMAs instead of 3
// (18, 36, 6), at the unchanged 48 KiB of LDS.  Same products as the three-segment walk, fp32 sums in another order (per K tile: hi.hi, hi.lo, lo.hi).
template <typename T>
__device__ __forceinline__ void mainloop_n64_x3(const GemmParams& p, int tm, int tn, char* lds, f32x16 (&acc)[2][1]) {
    constexpr int STAGE = DecGeo<64>::STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int row0_m = tm * BM, row0_n = tn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const int nk = p.seg[0].ktiles;                    // the three segments: the same K extent and leading dimensions (checked by the launcher)
    if (nk <= 0) return;
    uint32_t voA[4], voB[2];
    const uint32_t lda = (uint32_t)p.seg[0].lda_b, ldb = (uint32_t)p.seg[0].ldb_b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        voA[i] = (uint32_t)(row0_m + row) * lda + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        voB[i] = (uint32_t)(row0_n + row) * ldb + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    // segments as the launcher lists them: 0 = (h_hi, W_hi), 1 = (h_hi, W_lo), 2 = (h_lo, W_hi)
    const char *gAh = p.seg[0].A, *gBh = p.seg[0].Bt, *gAl = p.seg[2].A, *gBl = p.seg[1].Bt;
    auto dma_stage = [&](char* slot, const char* gA, const char* gB) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                             (__attribute__((address_space(3))) void*)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
    };
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const uint32_t offa = (wm * 64 + r) * BKB, offb = TILE_BYTES + (wn * 32 + r) * BKB;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
    dma_stage(lds, gAh, gBh);
    for (int i = 0; i < nk; ++i) {
        // ---- hi stage (slot 0): the lo stage of this K tile goes out first
        dma_stage(lds + STAGE, gAl, gBl);
        gAl += BKB; gBl += BKB;
        wait_vm<6>();
        __builtin_amdgcn_s_barrier();                  // the hi stage landed for every wave
        asm volatile("" ::: "memory");
        i32x4 fah[4][2], fbh[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fah[kk][0] = lds_read_b128(lbase + offa + so[kk]);
            fah[kk][1] = lds_read_b128_off4096(lbase + offa + so[kk]);
            fbh[kk] = lds_read_b128(lbase + offb + so[kk]);
        }
#define DAE_N64X_HI(KK, CNT)                                     \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<T>::run(fah[KK][0], fbh[KK], acc[0][0]);                 \
    Mma<T>::run(fah[KK][1], fbh[KK], acc[1][0]);
        DAE_N64X_HI(0, 9)
        DAE_N64X_HI(1, 6)
        DAE_N64X_HI(2, 3)
        DAE_N64X_HI(3, 0)
#undef DAE_N64X_HI
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // every wave has read slot 0: the next K tile's hi stage may refill it
        asm volatile("" ::: "memory");
        // ---- lo stage (slot 1)
        if (i + 1 < nk) {
            gAh += BKB; gBh += BKB;
            dma_stage(lds, gAh, gBh);
            wait_vm<6>();
        } else {
            wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();                  // the lo stage landed for every wave
        asm volatile("" ::: "memory");
        i32x4 fal[4][2], fbl[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fal[kk][0] = lds_read_b128(lbase + STAGE + offa + so[kk]);
            fal[kk][1] = lds_read_b128_off4096(lbase + STAGE + offa + so[kk]);
            fbl[kk] = lds_read_b128(lbase + STAGE + offb + so[kk]);
        }
#define DAE_N64X_LO(KK, CNT)                                     \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<T>::run(fah[KK][0], fbl[KK], acc[0][0]);                 \
    Mma<T>::run(fah[KK][1], fbl[KK], acc[1][0]);                 \
    Mma<T>::run(fal[KK][0], fbh[KK], acc[0][0]);                 \
    Mma<T>::run(fal[KK][1], fbh[KK], acc[1][0]);
        DAE_N64X_LO(0, 9)
        DAE_N64X_LO(1, 6)
        DAE_N64X_LO(2, 3)
        DAE_N64X_LO(3, 0)
#undef DAE_N64X_LO
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // every wave has read slot 1
        asm volatile("" ::: "memory");
    }
}

// Two-term form (the split modes that keep only W as hi + lo in the decode -- f16x2d, f16x2: z2 = h.W_hi^T + h.W_lo^T) with the same register carry: the hi
// stage holds (h, W_hi)[k], the lo stage ONLY the 8 KiB W_lo[k] tile, multiplied with the h fragments still in registers (32 VGPRs).  Per K tile and wave
// 8 LDS-DMA pieces and 16 fragment reads for the 16 MFMAs instead of 12 and 24 -- what mainloop_n64_pair buys, without its 64 KiB of LDS (three
// workgroups per CU stay).  Same products as the two-segment walk, (hi, lo) interleaved per K tile.
template <typename T>
__device__ __forceinline__ void mainloop_n64_c2(const GemmParams& p, int tm, int tn, char* lds, f32x16 (&acc)[2][1]) {
    constexpr int STAGE = DecGeo<64>::STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int row0_m = tm * BM, row0_n = tn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const int nk = p.seg[0].ktiles;                    // both segments: the same A operand, K extent and leading dimensions (checked by the launcher)
    if (nk <= 0) return;
    uint32_t voA[4], voB[2];
    const uint32_t lda = (uint32_t)p.seg[0].lda_b, ldb = (uint32_t)p.seg[0].ldb_b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        voA[i] = (uint32_t)(row0_m + row) * lda + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        voB[i] = (uint32_t)(row0_n + row) * ldb + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const char *gA = p.seg[0].A, *gBh = p.seg[0].Bt, *gBl = p.seg[1].Bt;
    auto dma_hi = [&](char* slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                             (__attribute__((address_space(3))) void*)(slot + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gBh + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
        gA += BKB; gBh += BKB;
    };
    auto dma_lo = [&](char* slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gBl + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
        gBl += BKB;
    };
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const uint32_t offa = (wm * 64 + r) * BKB, offb = TILE_BYTES + (wn * 32 + r) * BKB;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
    dma_hi(lds);
    for (int i = 0; i < nk; ++i) {
        // ---- hi stage (slot 0); the W_lo tile of this K tile goes out first (slot 1, B region)
        dma_lo(lds + STAGE);
        wait_vm<2>();
        __builtin_amdgcn_s_barrier();                  // the hi stage landed for every wave
        asm volatile("" ::: "memory");
        i32x4 fa[4][2], fbh[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[kk][0] = lds_read_b128(lbase + offa + so[kk]);
            fa[kk][1] = lds_read_b128_off4096(lbase + offa + so[kk]);
            fbh[kk] = lds_read_b128(lbase + offb + so[kk]);
        }
#define DAE_N64C_HI(KK, CNT)                                     \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<T>::run(fa[KK][0], fbh[KK], acc[0][0]);                  \
    Mma<T>::run(fa[KK][1], fbh[KK], acc[1][0]);
        DAE_N64C_HI(0, 9)
        DAE_N64C_HI(1, 6)
        DAE_N64C_HI(2, 3)
        DAE_N64C_HI(3, 0)
#undef DAE_N64C_HI
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // every wave has read slot 0: the next K tile's hi stage may refill it
        asm volatile("" ::: "memory");
        // ---- lo stage: W_lo[k] against the h fragments in registers
        if (i + 1 < nk) { dma_hi(lds); wait_vm<6>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                  // the W_lo tile landed for every wave
        asm volatile("" ::: "memory");
        i32x4 fbl[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fbl[kk] = lds_read_b128(lbase + STAGE + offb + so[kk]);
#define DAE_N64C_LO(KK, CNT)                                     \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<T>::run(fa[KK][0], fbl[KK], acc[0][0]);                  \
    Mma<T>::run(fa[KK][1], fbl[KK], acc[1][0]);
        DAE_N64C_LO(0, 3)
        DAE_N64C_LO(1, 2)
        DAE_N64C_LO(2, 1)
        DAE_N64C_LO(3, 0)
#undef DAE_N64C_LO
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // every wave has read the W_lo tile
        asm volatile("" ::: "memory");
    }
}

constexpr float CE_FAST_ZMAX = 14.0f;               // sigmoid(14) = 1 - 8.3e-7: five fp32 ulps from saturation
constexpr int DECODE_NST = 2;
template <typename T, int LOSS, int ACT, bool XBITS = false, int BN_T = 128, bool RES = false, bool PAIRD = false, bool X3 = false, bool C2 = false>   // RES (split-bf16 mode): also the lo images of delta2 / delta2^T; PAIRD: mainloop_n64_pair; X3: mainloop_n64_x3; C2: mainloop_n64_c2
__global__ __launch_bounds__(GEMM_THREADS, PAIRD ? 2 : DecGeo<BN_T>::WG_PER_CU) void gemm_decode_loss(GemmParams p, DecodeEpi e) {
    static_assert(!PAIRD || (BN_T == 64 && sizeof(T) == 2 && !RES), "the paired K loop exists for the 64-column 16-bit kernel");
    static_assert(!X3 || (BN_T == 64 && sizeof(T) == 2 && !RES && !PAIRD), "the three-term K loop exists for the 64-column 16-bit kernel");
    static_assert(!C2 || (BN_T == 64 && sizeof(T) == 2 && !X3 && !PAIRD), "the two-term register-carry K loop exists for the 64-column 16-bit kernel");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using Geo = DecGeo<BN_T>;
    constexpr int NTB = Geo::NTB, WCOLS = Geo::WCOLS, P0 = Geo::P0, P1 = Geo::P1;
    constexpr int WPR = BN_T / 32;                     // words of the x bit tile per row
    constexpr bool STAGED = (sizeof(T) == 2);
    static_assert(!XBITS || STAGED, "the bit image of x is a bf16-mode operand");
    static_assert(BN_T == 128 || STAGED, "the 64-column tile is a bf16-mode kernel");
    static_assert(!RES || STAGED, "lo images exist for bf16 operands only");
    constexpr bool IS_COS = (LOSS == DAE_LOSS_COSINE);
    if (e.sym_G && (int)blockIdx.x >= e.sym_first) {   // rider workgroups: the symmetrised triplet gradient (see DecodeEpi)
        const int t = (int)blockIdx.x - e.sym_first, nt = e.sym_Bp / 64;
        sym_scale_tile<T>(e.sym_G, e.sym_B, e.sym_Bp, e.sym_scalars, reinterpret_cast<T*>(e.sym_Gs), t % nt, t / nt,
                          reinterpret_cast<float(*)[65]>(lds), STAGED ? e.op_scale : 1.f);
        return;
    }
    int tm, tn, split, kt0, kt1;
    if (!block_to_tile(p, tm, tn, split, kt0, kt1)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 5, c = lane & 31;
    const T* X = reinterpret_cast<const T*>(e.x);
    T* D2 = reinterpret_cast<T*>(e.delta2);
    T* D2T = reinterpret_cast<T*>(e.delta2_t);
    const bool pass1 = IS_COS && e.cos_pass == 1;

    // ---- prefetch the clean-input tile (registers now, LDS after the K loop) ----
    // XBITS (binary input): the tile is 128 rows x BN_T/32 words of the gather's bit image instead of 128 x BN_T bf16
    constexpr int XCH = BN_T / 16;                     // 16-byte chunks per thread of the [128][BN_T] bf16 tile
    i32x4 xr[XCH];
    uint32_t xb[WPR / 2] = {};
    if constexpr (XBITS) {
#pragma unroll
        for (int w = 0; w < WPR / 2; ++w)
            xb[w] = e.x_bits[(int64_t)(tm * BM + (tid >> 1)) * e.ldxb + tn * WPR + (tid & 1) * (WPR / 2) + w];
    } else if constexpr (STAGED) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int ch = tid + GEMM_THREADS * i;
            const int row = ch / (BN_T / 8), c16 = ch % (BN_T / 8);
            xr[i] = *reinterpret_cast<const i32x4*>(X + (int64_t)(tm * BM + row) * e.ldx + tn * BN_T + c16 * 8);
        }
    }

    f32x16 acc[2][NTB];
    // 32-row blocks of this tile that hold a real batch row (1..4); block (wm, mt) of this wave is 2 wm + mt
    const int vblk = e.no_pad_skip ? 4 : min(4, (e.B - tm * BM + 31) >> 5);
    const bool vb[2] = {2 * wm < vblk, 2 * wm + 1 < vblk};
    // cosine, second pass: the accumulators of the statistics pass come back from memory (DecodeEpi::z_io) instead of a second walk over K -- the decode
    // GEMM of this loss runs ONCE per step.
    const bool z_load = IS_COS && e.z_mode == 2 && e.z_io;
    if (z_load) {
        // (the buffer is private to these two passes: it holds the accumulator registers themselves, tile by tile and wave by wave, as 16-byte pieces of
        //  consecutive lanes -- 8 NTB fully coalesced 1-KiB accesses per wave instead of 32 NTB dword ones on row-major logits)
        const f32x4* zp = reinterpret_cast<const f32x4*>(e.z_io) + ((int64_t)(tm * (e.Fp / BN_T) + tn) * 4 + wave) * (2 * NTB * 4 * 64) + lane;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = zp[((mt * NTB + nt) * 4 + q) * 64];
                    acc[mt][nt][4 * q] = v[0]; acc[mt][nt][4 * q + 1] = v[1]; acc[mt][nt][4 * q + 2] = v[2]; acc[mt][nt][4 * q + 3] = v[3];
                }
    } else {
    if constexpr (BN_T == 128) gemm_mainloop<T, DECODE_NST>(p, tm, tn, kt0, kt1, lds, acc);
    else if constexpr (PAIRD) mainloop_n64_pair<T>(p, tm, tn, lds, acc);
    else if constexpr (X3) mainloop_n64_x3<T>(p, tm, tn, lds, acc);
    else if constexpr (C2) mainloop_n64_c2<T>(p, tm, tn, lds, acc);
    else mainloop_n64<T>(p, tm, tn, lds, acc);
    }
    if (IS_COS && e.z_mode == 1 && e.z_io) {           // statistics pass: park the accumulators for the final pass (128-byte row segments per half wave)
        f32x4* zp = reinterpret_cast<f32x4*>(e.z_io) + ((int64_t)(tm * (e.Fp / BN_T) + tn) * 4 + wave) * (2 * NTB * 4 * 64) + lane;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                    zp[((mt * NTB + nt) * 4 + q) * 64] = v;
                }
    }
    // (the K loop multiplies the padding blocks too: branching around MFMAs would put them in basic blocks of their own, out of reach of the static check
    //  of the hand-placed LDS waits, tools/check_gemm_asm.py -- measured worth < 1 % of the step; the EPILOGUE below skips their loss evaluation)
    __syncthreads();                                   // every wave is done with the K-loop stages

    char* R0 = lds;                                    // x tile, overwritten in place by delta2   [128][P0]
    char* R1 = lds + Geo::R0_BYTES;                    // delta2^T tile                             [BN_T][P1]
    float* aux = reinterpret_cast<float*>(lds + Geo::AUX_OFF);
    float* cw_l = aux;                                 // [128] row weights
    float* bv_l = aux + 128;                           // [BN_T] visible bias
    float* rowsum_l = aux + 256;                       // [2 (wn)][128]
    float* colsum_l = aux + 512;                       // [2 (wm)][BN_T]
    float* inx_l = aux + 768;                          // [128] 1/|x|            (cosine)
    float* cyy_l = aux + 896;                          // [128] sum y^2          (cosine pass 2)
    float* cxy_l = aux + 1024;                         // [128] sum xhat.y       (cosine pass 2)
    float* pyy_l = aux + 1152;                         // [2][128] partial sum y^2   (cosine pass 1)
    float* pxy_l = aux + 1408;                         // [2][128] partial sum xhat.y
    uint32_t* xb_l = reinterpret_cast<uint32_t*>(aux + 1664);   // [128][WPR] bit image of the clean-input tile (XBITS)

    if constexpr (XBITS) {
#pragma unroll
        for (int w = 0; w < WPR / 2; ++w) xb_l[(tid >> 1) * WPR + (tid & 1) * (WPR / 2) + w] = xb[w];
    } else if constexpr (STAGED) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int ch = tid + GEMM_THREADS * i;
            const int row = ch / (BN_T / 8), c16 = ch % (BN_T / 8);
            *reinterpret_cast<i32x4*>(R0 + row * P0 + c16 * 16) = xr[i];
        }
    }
    if (tid < 128) {
        const int row = tm * BM + tid, col = tn * BN_T + tid;
        cw_l[tid] = e.cw[row];                         // zero beyond B by construction
        if (tid < BN_T) bv_l[tid] = col < e.F ? e.bv[col] : 0.f;
        if constexpr (IS_COS) {
            inx_l[tid] = rsqrtf(fmaxf(e.cos_stats[row], 1e-12f));
            cyy_l[tid] = e.cos_pass == 2 ? e.cos_stats[e.Bp + row] : 0.f;
            cxy_l[tid] = e.cos_pass == 2 ? e.cos_stats[2 * e.Bp + row] : 0.f;
        }
    }
    __syncthreads();

    const int lcol0 = wn * WCOLS + c;                  // local column of nt = 0
    const int lrow0 = wm * 64 + 4 * g;                 // local row of (mt = 0, r = 0)
    const float eps = 1e-16f;
    float colsum[NTB];
    float cm[NTB], bvv[NTB];                           // column mask as a multiplier: padded features contribute nothing
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt) {
        colsum[nt] = 0.f;
        cm[nt] = (tn * BN_T + lcol0 + nt * 32) < e.F ? 1.f : 0.f;
        bvv[nt] = bv_l[lcol0 + nt * 32];
    }
    // per-lane base addresses; everything else is a compile-time offset
    char* r0_lane = R0 + lrow0 * P0 + lcol0 * 2;
    char* r1_lane = R1 + lcol0 * P1 + lrow0 * 2;
    const T* x_lane = X + (int64_t)(tm * BM + lrow0) * e.ldx + tn * BN_T + lcol0;
    const bf16_t* x2_lane = (RES && e.x2) ? reinterpret_cast<const bf16_t*>(e.x2) + (int64_t)(tm * BM + lrow0) * e.ldx + tn * BN_T + lcol0 : nullptr;
    T* d2_lane = D2 ? D2 + (int64_t)(tm * BM + lrow0) * e.ldd + tn * BN_T + lcol0 : nullptr;
    T* d2t_lane = D2T ? D2T + (int64_t)(tn * BN_T + lcol0) * e.lddt + tm * BM + lrow0 : nullptr;

    // static (compile-time) accumulator indexing: a runtime-indexed f32x16 would be demoted to scratch
    // FAST (cross_entropy + sigmoid, every |z| of this wave < CE_FAST_ZMAX): the exact-math identities
    //   -log y = softplus(-z), -log(1-y) = softplus(z), dL/dy * y(1-y) = y - x
    // replace the reference's  log(y+1e-16), log((1-y)+1e-16), 1/(y+1e-16), 1/((1-y)+1e-16)  -- 3 transcendentals per
    // element instead of 6.  In that range the 1e-16 guards are below fp32 resolution (y, 1-y > 3e-7), so the two forms
    // differ only by fp32 rounding (the softplus form is the more accurate one).  Waves holding a saturated logit take
    // the reference-literal path, which reproduces TF's fp32 behaviour there (y rounds to 1, log(1e-16) = -36.84).
    const bool want_rows = e.rowloss_part != nullptr;
    const float osc = e.op_scale;
    float wl_acc = 0.f;                                // this lane's share of sum_i cw_i * loss_if
    uint32_t resv[RES ? 8 : 1][NTB][2];                // RES: packed bf16(d2 - bf16(d2)) of this lane's elements, block (mt, r4) at index mt * 4 + r4
    auto epi_block = [&](auto MT, auto R4, auto FASTV) {
        constexpr int mt = decltype(MT)::value, r4 = decltype(R4)::value;
        constexpr bool FAST = decltype(FASTV)::value;
        constexpr int rloc = mt * 32 + 8 * r4;         // local row offset of q = 0 relative to lrow0
        float d2v[NTB][4];
        float xin[4][NTB];
        float xlo[RES ? 4 : 1][RES ? NTB : 1];
        if constexpr (RES) {                           // lo image of the clean rows (NULL for bf16-exact data): straight from memory
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt)
                    xlo[q][nt] = x2_lane ? bf2f(x2_lane[(int64_t)(rloc + q) * e.ldx + nt * 32]) : 0.f;
        }
        if constexpr (!STAGED) {                       // fp32: batch the global loads of this block
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt) xin[q][nt] = Elem<T>::to(x_lane[(int64_t)(rloc + q) * e.ldx + nt * 32]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            constexpr int rbase = r4 * 4;
            const int r = rbase + q;
            const int lrow = lrow0 + rloc + q;
            const float cwi = cw_l[lrow];
            float rl = 0.f, s_yy = 0.f, s_xy = 0.f;
            float inx = 0.f, cs_yy = 0.f, cs_xy = 0.f;
            if constexpr (IS_COS) { inx = inx_l[lrow]; cs_yy = cyy_l[lrow]; cs_xy = cxy_l[lrow]; }
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) {
                const float z = acc[mt][nt][r] + bvv[nt];
                float x;
                if constexpr (XBITS) x = (float)((xb_l[(lrow0 + rloc + q) * WPR + wn * NTB + nt] >> c) & 1u);
                else if constexpr (STAGED) {
                    x = bf2f(*reinterpret_cast<const bf16_t*>(r0_lane + (rloc + q) * P0 + nt * 64));
                    if constexpr (RES) x += xlo[q][nt];         // valued input in split-bf16 mode: x = hi + lo
                } else x = xin[q][nt];
                float l = 0.f, dy = 0.f;
                if constexpr (FAST) {
                    const float en = __builtin_amdgcn_exp2f(-fabsf(z) * kLog2e);          // exp(-|z|)
                    const float op = 1.0f + en;
                    const float rr = __builtin_amdgcn_rcpf(op);
                    const float yv = z >= 0.f ? rr : en * rr;
                    l = kLn2 * __builtin_amdgcn_logf(op) + fmaxf(z, 0.f) - x * z;
                    const float d2 = (cwi * cm[nt]) * (yv - x);
                    rl += cm[nt] * l;
                    colsum[nt] += d2;
                    if constexpr (STAGED) {
                        const float d2s = sat16(d2 * osc);              // the 16-bit images hold op_scale * delta2
                        d2v[nt][q] = d2s;
                        *reinterpret_cast<bf16_t*>(r0_lane + (rloc + q) * P0 + nt * 64) = f2bf_hw(d2s);
                    } else {
                        d2v[nt][q] = d2;
                        if (d2_lane) d2_lane[(int64_t)(rloc + q) * e.ldd + nt * 32] = Elem<T>::from(d2);
                    }
                    continue;
                }
                const float y = act_fwd<ACT>(z);
                if constexpr (LOSS == DAE_LOSS_CROSS_ENTROPY) {
                    const float a = y + eps, b = (1.0f - y) + eps;          // reference op order: (1.-y)+1e-16
                    const float la = __builtin_amdgcn_logf(a), lb = __builtin_amdgcn_logf(b);   // log2; a, b >= 1e-16 (normal)
                    l = -kLn2 * (x * la + (1.0f - x) * lb);
                    dy = (1.0f - x) * __builtin_amdgcn_rcpf(b) - x * __builtin_amdgcn_rcpf(a);  // the two logs are differentiated separately
                } else if constexpr (LOSS == DAE_LOSS_MEAN_SQUARED) {
                    const float d = x - y;
                    l = d * d;
                    dy = -2.0f * d;
                } else {
                    const float xh = x * inx;
                    if (pass1) {
                        s_yy += cm[nt] * y * y;
                        s_xy += cm[nt] * xh * y;
                    } else {
                        const float big = cs_yy >= 1e-12f ? 1.f : 0.f;      // tf.maximum routes grad to sum y^2 iff >= eps
                        const float s = rsqrtf(fmaxf(cs_yy, 1e-12f));
                        dy = -(xh * s - big * cs_xy * s * s * s * y);
                    }
                }
                const float d2 = pass1 ? 0.f : (cwi * cm[nt]) * dy * act_bwd<ACT>(y);   // cw is 0 on padded rows
                rl += cm[nt] * l;
                colsum[nt] += d2;
                if constexpr (STAGED) {
                    const float d2s = sat16(d2 * osc);
                    d2v[nt][q] = d2s;
                    *reinterpret_cast<bf16_t*>(r0_lane + (rloc + q) * P0 + nt * 64) = f2bf_hw(d2s);
                } else {
                    d2v[nt][q] = d2;
                    if (d2_lane && !pass1) d2_lane[(int64_t)(rloc + q) * e.ldd + nt * 32] = Elem<T>::from(d2);
                }
            }
            // wavefront (DPP) sum over the 32 lanes that share this row; lanes 16..31 / 48..63 hold it
            if constexpr (IS_COS) {
                if (pass1) {
                    s_yy = half32_sum_hi(s_yy); s_xy = half32_sum_hi(s_xy);
                    if (c == 31) { pyy_l[wn * 128 + lrow] = s_yy; pxy_l[wn * 128 + lrow] = s_xy; }
                }
            } else {
                wl_acc += cwi * rl;
                if (want_rows) {
                    rl = half32_sum_hi(rl);
                    if (c == 31) rowsum_l[wn * 128 + lrow] = rl;
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) {
            if constexpr (STAGED) {
                if (D2T) {                                  // (NULL: the dW kernel reads delta2 row-major, no transposed image -- gemm_dw_pc<TRA>)
                    uint2 v;
                    v.x = f2bf_pack_hw(d2v[nt][0], d2v[nt][1]);
                    v.y = f2bf_pack_hw(d2v[nt][2], d2v[nt][3]);
                    *reinterpret_cast<uint2*>(r1_lane + nt * 32 * P1 + rloc * 2) = v;
                }
                if constexpr (RES) {
                    resv[mt * 4 + r4][nt][0] = bf_residual_pack_hw(d2v[nt][0], d2v[nt][1]);
                    resv[mt * 4 + r4][nt][1] = bf_residual_pack_hw(d2v[nt][2], d2v[nt][3]);
                }
            } else {
                if (d2t_lane && !pass1)
                    store4<T>(d2t_lane + (int64_t)nt * 32 * e.lddt + rloc, d2v[nt][0], d2v[nt][1], d2v[nt][2], d2v[nt][3]);
            }
        }
    };
    // a 32-row block of pure padding (rows >= B): delta2 = 0 without evaluating the loss (cw is 0 there, so the evaluated form stores the same zeros)
    auto zero_block = [&](auto MT) {
        constexpr int mt = decltype(MT)::value;
        if constexpr (STAGED) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rloc = mt * 32 + 8 * r4;
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<bf16_t*>(r0_lane + (rloc + q) * P0 + nt * 64) = (bf16_t)0;
                    uint2 z; z.x = 0u; z.y = 0u;
                    if (D2T) *reinterpret_cast<uint2*>(r1_lane + nt * 32 * P1 + rloc * 2) = z;
                    if constexpr (RES) { resv[mt * 4 + r4][nt][0] = 0u; resv[mt * 4 + r4][nt][1] = 0u; }
                }
            }
        } else {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt) {
                    const int rloc = mt * 32 + 8 * r4;
                    if (!pass1) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (d2_lane) d2_lane[(int64_t)(rloc + q) * e.ldd + nt * 32] = Elem<T>::from(0.f);
                        if (d2t_lane) store4<T>(d2t_lane + (int64_t)nt * 32 * e.lddt + rloc, 0.f, 0.f, 0.f, 0.f);
                    }
                }
        }
    };
#define DAE_EPI_ROWS(MTV, FV)                                                                                              \
    if (vb[MTV]) {                                                                                                         \
    epi_block(std::integral_constant<int, MTV>{}, std::integral_constant<int, 0>{}, std::integral_constant<bool, FV>{});   \
    epi_block(std::integral_constant<int, MTV>{}, std::integral_constant<int, 1>{}, std::integral_constant<bool, FV>{});   \
    epi_block(std::integral_constant<int, MTV>{}, std::integral_constant<int, 2>{}, std::integral_constant<bool, FV>{});   \
    epi_block(std::integral_constant<int, MTV>{}, std::integral_constant<int, 3>{}, std::integral_constant<bool, FV>{});   \
    } else zero_block(std::integral_constant<int, MTV>{});
    bool fast = false;
    if constexpr (LOSS == DAE_LOSS_CROSS_ENTROPY && ACT == DAE_ACT_SIGMOID) {
        float zmax = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) zmax = fmaxf(zmax, fabsf(acc[mt][nt][r] + bvv[nt]));
        fast = __builtin_amdgcn_ballot_w64(!(zmax < CE_FAST_ZMAX)) == 0ull && !e.ce_literal;   // NaN logits take the literal path
    }
    if (fast) {
        if constexpr (LOSS == DAE_LOSS_CROSS_ENTROPY && ACT == DAE_ACT_SIGMOID) {
            DAE_EPI_ROWS(0, true)
            DAE_EPI_ROWS(1, true)
        }
    } else {
        DAE_EPI_ROWS(0, false)
        DAE_EPI_ROWS(1, false)
    }
#undef DAE_EPI_ROWS
    if (!pass1) {                                       // column sums: rows of g = 0 and g = 1, then one lane per column
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt) {
            const float v = colsum[nt] + __shfl_xor(colsum[nt], 32, 64);
            if (g == 0) colsum_l[wm * BN_T + lcol0 + nt * 32] = v;
        }
    }
    __syncthreads();

    // ---- leave the CU: coalesced tiles and per-wave partial sums ----
    if constexpr (STAGED) {
        if (!pass1) {
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const int ch = tid + GEMM_THREADS * i;
                if (D2) {
                    const int row = ch / (BN_T / 8), c16 = ch % (BN_T / 8);
                    *reinterpret_cast<i32x4*>(D2 + (int64_t)(tm * BM + row) * e.ldd + tn * BN_T + c16 * 8) =
                        *reinterpret_cast<const i32x4*>(R0 + row * P0 + c16 * 16);
                }
                if (D2T) {
                    const int row = ch >> 4, c16 = ch & 15;
                    *reinterpret_cast<i32x4*>(D2T + (int64_t)(tn * BN_T + row) * e.lddt + tm * BM + c16 * 8) =
                        *reinterpret_cast<const i32x4*>(R1 + row * P1 + c16 * 16);
                }
            }
            if constexpr (RES) {                        // second round through the same two staging tiles: the lo images
                __syncthreads();                        // every piece of the hi tiles has been read
#pragma unroll
                for (int blk = 0; blk < 8; ++blk) {
                    const int rloc = (blk >> 2) * 32 + 8 * (blk & 3);
#pragma unroll
                    for (int nt = 0; nt < NTB; ++nt) {
                        const uint32_t a = resv[blk][nt][0], b = resv[blk][nt][1];
                        *reinterpret_cast<bf16_t*>(r0_lane + (rloc + 0) * P0 + nt * 64) = (bf16_t)(a & 0xffffu);
                        *reinterpret_cast<bf16_t*>(r0_lane + (rloc + 1) * P0 + nt * 64) = (bf16_t)(a >> 16);
                        *reinterpret_cast<bf16_t*>(r0_lane + (rloc + 2) * P0 + nt * 64) = (bf16_t)(b & 0xffffu);
                        *reinterpret_cast<bf16_t*>(r0_lane + (rloc + 3) * P0 + nt * 64) = (bf16_t)(b >> 16);
                        uint2 v; v.x = a; v.y = b;
                        *reinterpret_cast<uint2*>(r1_lane + nt * 32 * P1 + rloc * 2) = v;
                    }
                }
                __syncthreads();
                T* D2b = reinterpret_cast<T*>(e.delta2_2);
                T* D2Tb = reinterpret_cast<T*>(e.delta2_t2);
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const int ch = tid + GEMM_THREADS * i;
                    if (D2b) {
                        const int row = ch / (BN_T / 8), c16 = ch % (BN_T / 8);
                        *reinterpret_cast<i32x4*>(D2b + (int64_t)(tm * BM + row) * e.ldd + tn * BN_T + c16 * 8) =
                            *reinterpret_cast<const i32x4*>(R0 + row * P0 + c16 * 16);
                    }
                    if (D2Tb) {
                        const int row = ch >> 4, c16 = ch & 15;
                        *reinterpret_cast<i32x4*>(D2Tb + (int64_t)(tn * BN_T + row) * e.lddt + tm * BM + c16 * 8) =
                            *reinterpret_cast<const i32x4*>(R1 + row * P1 + c16 * 16);
                    }
                }
            }
        }
    }
    {
        const int w = tid >> 7, k = tid & 127;          // 256 threads = 2 partial rows x 128
        const bool rowok = (tm * BM + k) < e.B;
        if constexpr (IS_COS) {
            if (pass1) {
                e.cos_part[(int64_t)(tn * 2 + w) * e.Bp + tm * BM + k] = rowok ? pyy_l[w * 128 + k] : 0.f;
                e.cos_part[(int64_t)(2 * p.tiles_n + tn * 2 + w) * e.Bp + tm * BM + k] = rowok ? pxy_l[w * 128 + k] : 0.f;
            }
        } else {
            if (e.rowloss_part) e.rowloss_part[(int64_t)(tn * 2 + w) * e.Bp + tm * BM + k] = rowok ? rowsum_l[w * 128 + k] : 0.f;
        }
        if (e.dbv_part && !pass1 && tid < 2 * BN_T) {
            const int w2 = tid / BN_T, k2 = tid % BN_T;
            e.dbv_part[(int64_t)(tm * 2 + w2) * e.Fp + tn * BN_T + k2] = colsum_l[w2 * BN_T + k2];
        }
    }
    if constexpr (!IS_COS) {
        if (e.tile_part) {                              // this tile's share of sum_i cw_i * rowloss_i (4 waves, fixed order)
            const float v = wave64_sum_hi(wl_acc);
            if (lane == 63) pyy_l[wave] = v;            // pyy_l is unused outside cosine
        }
        __syncthreads();
        if (e.tile_part && tid == 0) e.tile_part[tm * p.tiles_n + tn] = (pyy_l[0] + pyy_l[1]) + (pyy_l[2] + pyy_l[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// A-stationary persistent decode (round 6; 16-bit modes, K = Hp <= 512, every K segment over the SAME h -- f16x2's (h, W_hi) (h, W_lo), plain bf16 / f16).
//
// gemm_decode_loss<.., 64> is bound by neither the MFMA pipe nor HBM at c2: per K tile a 128 x 64 workgroup moves 24 KiB through LDS-DMA and reads 48 KiB
// of fragments for 32 MFMAs, holds ONE K tile in flight (48 KiB of LDS = two stages), and a launch is 1106 one-tile workgroups in 1.44 rounds of the chip's 768
// slots (tools/decode_quant_probe.sh: t = 17 us + 24 ns per tile).  Every phase of it is a latency chain.  Here
//   * h never stays in LDS: wave w of a workgroup owns rows [32 w, +32) of a 128-row panel and keeps their fragments over the WHOLE K in registers
//     (8 K tiles x 4 k-steps x 16 B = 128 VGPRs), filled once per panel through the same ring the W tiles use; both W terms of the split modes multiply
//     the same registers;
//   * only W streams afterwards: one 64-row B tile (8 KiB) per K tile through a 7-slot LDS-DMA ring -- six K tiles in flight per workgroup instead of one --
//     at ONE barrier per K tile; a wave reads the whole B tile (8 ds_read_b128) for its 8 MFMAs: 40 KiB of LDS traffic per K tile instead of 72;
//   * workgroups are persistent, two per CU; each walks a contiguous run of tiles of ITS XCD's column band (the band's W rows, 2.5 MB hi + lo, stay in that
//     XCD's L2).  The ring runs across tile boundaries: while a tile's loss is evaluated the next tile's first six B tiles are already on their way;
//   * the epilogue is wave-local: a wave's 32 rows are its own, so the clean-input bits, the row weights, the staged delta2 rows and their coalesced
//     stores need no workgroup barrier (one remains, for the column sums of the bias gradient); delta2^T leaves straight from the accumulator layout
//     (8-byte pieces: four consecutive batch rows of one feature), so no second staging tile exists;
//   * the Gs rider tiles are done by the same workgroups before their first slots land (no extra workgroups queueing for a slot).
// The arithmetic is gemm_decode_loss's (same FAST / literal split, same op_scale, same partial-sum layouts; the column / row partial sums are added in the
// order of the 4 x 1 wave layout).  Reference: autoencoder.py:395-415 (decode), triplet_loss_utils.py:262-277 (weighted_loss).
// ------------------------------------------------------------------------------------------------
// workgroup barrier that orders LDS accesses only: __syncthreads() also waits vmcnt(0), i.e. for every LDS-DMA slot and global store still in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct DecAst {
    static constexpr int NST = 6;                          // ring slots (7 would be 79 KiB per workgroup: measured ONE resident workgroup per CU then)
    static constexpr int SLOT = 64 * BKB;                  // one slot: 64 rows x 128 B = 8 KiB (a B tile of W rows, or half of an A tile of h rows)
    static constexpr int RING = NST * SLOT;
    static constexpr int MAXKT = 8;                        // K tiles whose A fragments a wave holds (Hp <= 512 in 16-bit elements)
    static constexpr int P0 = 64 * 2 + 16;                 // staged row pitch of a wave's [32][64] delta2 rows (16-bit + 16 B pad)
    static constexpr int R0W = 32 * P0;                    // per wave
    // floats behind the four R0W blocks: per wave [32] cw, [32] rowsum / sum y^2, [32] sum xhat.y, [32] 1/|x|, [32] cyy, [32] cxy, [32][2] x bits; shared [4][64] colsum, [4] loss shares
    static constexpr int WAVE_FLOATS = 6 * 32 + 64;
    static constexpr int AUX_FLOATS = 4 * WAVE_FLOATS + 4 * 64 + 4;
    static constexpr int EPI_BYTES = 4 * R0W + AUX_FLOATS * 4;
    static constexpr int LDS_BYTES = RING + EPI_BYTES;
};
static_assert(2 * DecAst::LDS_BYTES <= 160 * 1024, "two persistent decode workgroups per CU");
static_assert(DecAst::RING >= 64 * 65 * 4, "the Gs rider tile fits the ring");
__device__ __forceinline__ void wait_vm_2n(int n) {        // vmcnt(2 n): the pieces of n younger ring slots (2 per wave each) may stay in flight
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<2>(); break;
        case 2: wait_vm<4>(); break;
        case 3: wait_vm<6>(); break;
        case 4: wait_vm<8>(); break;
        case 5: wait_vm<10>(); break;
        default: wait_vm<12>(); break;
    }
}

template <int LOSS, int ACT, bool XBITS>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_decode_ast(GemmParams p, DecodeEpi e) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using T = bf16_t;
    constexpr int BN_T = 64, NT = 2, P0 = DecAst::P0, NST = DecAst::NST, SLOT = DecAst::SLOT;
    constexpr bool IS_COS = (LOSS == DAE_LOSS_COSINE);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, c = lane & 31;
    char* epi = lds + DecAst::RING;
    // timing probe (DecodeEpi::dbg & 64): workgroup b leaves 100 MHz timestamps in dbv_part[b * 32 ..] (as uint64) instead of the bias-gradient partials
    int n_stamp = 0;
    auto stamp = [&]() {
        if ((e.dbg & 64) && threadIdx.x == 0 && n_stamp < 16)
            reinterpret_cast<unsigned long long*>(e.dbv_part)[(int64_t)blockIdx.x * 16 + n_stamp] = __builtin_amdgcn_s_memrealtime();
        ++n_stamp;
    };
    stamp();

    // ---- this workgroup's run of tiles: XCD x = b % nx owns the column tiles [c0, c1); its workgroups split the band's tiles_m * nc tiles, panel-major ----
    const int nwg = (int)gridDim.x, b = (int)blockIdx.x;
    const int nx = nwg < 8 ? nwg : 8;
    const int x = b % nx, j = b / nx, J = (nwg - x + nx - 1) / nx;
    const int c0 = (p.tiles_n * x) / nx, c1 = (p.tiles_n * (x + 1)) / nx, nc = c1 - c0;
    const int n_x = p.tiles_m * nc;
    const int i0 = (int)(((int64_t)n_x * j) / J), i1 = (int)(((int64_t)n_x * (j + 1)) / J);
    const int ntiles = i1 - i0;
    const int nkt = p.seg[0].ktiles, nseg = p.nseg;                          // K tiles per segment (<= 8); segments share h

    // ---- Gs rider tiles (DecodeEpi::sym_*) first: the ring is still empty ----
    if (e.sym_G) {
        const int nt64 = e.sym_Bp / 64;
        for (int t = b; t < nt64 * nt64; t += nwg) {
            sym_scale_tile<T>(e.sym_G, e.sym_B, e.sym_Bp, e.sym_scalars, reinterpret_cast<T*>(e.sym_Gs), t % nt64, t / nt64,
                              reinterpret_cast<float(*)[65]>(lds), e.op_scale);
            __syncthreads();
        }
    }
    if (ntiles <= 0) return;

    // ---- producer: the slot stream, in RUNS of nkt slots (K tiles 0 .. nkt-1 of one 64-row block).  Per tile: [h rows [0, 64) | h rows [64, 128)] when the tile
    // opens a new panel, then one run of W rows per K segment.  Kept small on purpose (one call site, pointer bumps; the run switch is the only slow path):
    // an earlier form with the switch logic inlined at 24 sites of an unrolled K loop was 60 KB of code and spent 0.3 - 0.7 us per ring step fetching it.
    constexpr int L = NST - 1;                             // slots in flight ahead of the consumer
    const uint32_t lda = (uint32_t)p.seg[0].lda_b, ldb = (uint32_t)p.seg[0].ldb_b;
    uint32_t vrow[2], vsl[2];                              // this lane's row of a slot's two pieces and the (swizzled) 16-byte slot it fetches
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        vrow[i] = (uint32_t)((i * 4 + wave) * 8 + (lane >> 3));
        vsl[i] = (uint32_t)(((lane & 7) ^ ((vrow[i] >> 1) & 7)) << 4);
    }
    const int runs_per_tile = 2 + nseg;                    // run 0 / 1: the A halves (only when the panel changes), run 2 + s: segment s
    int p_it = 0, p_run = 0, p_left = 0, p_tm = -1, p_tn = 0;
    uint32_t ps_off = 0;                                   // ring byte offset of the next slot to fill
    const char* p_src = nullptr;
    uint32_t vo0 = 0, vo1 = 0;
    bool p_more = true;
    auto run_setup = [&]() {                               // p_run of tile p_it starts
        p_left = nkt;
        // (no dynamic index into the kernarg struct, no table of per-lane offsets: either would live in scratch, a ~1 us reload on the producer's path)
        const uint32_t ld = p_run < 2 ? lda : ldb;
        p_src = p_run < 2 ? p.seg[0].A + ((int64_t)p_tm * BM + p_run * 64) * lda : (p_run == 2 ? p.seg[0].Bt : p.seg[1].Bt) + (int64_t)p_tn * BN_T * ldb;
        vo0 = vrow[0] * ld + vsl[0];
        vo1 = vrow[1] * ld + vsl[1];
    };
    auto tile_open = [&]() {
        const int i = i0 + p_it;
        const int tm = i / nc;
        p_tn = c0 + i % nc;
        p_run = tm != p_tm ? 0 : 2;
        p_tm = tm;
        run_setup();
    };
    auto dma_one = [&]() {
        char* slot = lds + ps_off;
        if (!(e.dbg & 4)) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p_src + vo0),
                                             (__attribute__((address_space(3))) void*)(slot + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p_src + vo1),
                                             (__attribute__((address_space(3))) void*)(slot + (4 + wave) * 1024), 16, 0, 0);
        }
        p_src += BKB;
        ps_off += SLOT;
        if (ps_off == DecAst::RING) ps_off = 0;
        if (--p_left == 0) {                               // next run / next tile / end of the stream
            if (++p_run < runs_per_tile) run_setup();
            else if (++p_it < ntiles) tile_open();
            else p_more = false;
        }
    };
    tile_open();
#pragma nounroll
    for (int s = 0; s < L && p_more; ++s) dma_one();
    stamp();

    const T* X = reinterpret_cast<const T*>(e.x);
    T* D2 = reinterpret_cast<T*>(e.delta2);
    T* D2T = reinterpret_cast<T*>(e.delta2_t);
    const bool pass1 = IS_COS && e.cos_pass == 1;
    char* R0 = epi + wave * DecAst::R0W;               // this wave's rows: x tile, overwritten in place by delta2   [32][P0]
    float* auxw = reinterpret_cast<float*>(epi + 4 * DecAst::R0W) + wave * DecAst::WAVE_FLOATS;
    float* cw_l = auxw;                                // [32] row weights of this wave's rows
    float* rs_l = auxw + 32;                           // [32] row sums (loss), or sum y^2 (cosine pass 1)
    float* xy_l = auxw + 64;                           // [32] sum xhat.y (cosine pass 1)
    float* inx_l = auxw + 96;                          // [32] 1/|x|            (cosine)
    float* cyy_l = auxw + 128;                         // [32] sum y^2          (cosine pass 2)
    float* cxy_l = auxw + 160;                         // [32] sum xhat.y       (cosine pass 2)
    uint32_t* xb_l = reinterpret_cast<uint32_t*>(auxw + 192);   // [32][2] bit image of this wave's clean-input rows (XBITS)
    float* colsum_l = reinterpret_cast<float*>(epi + 4 * DecAst::R0W) + 4 * DecAst::WAVE_FLOATS;   // [4 (wave)][64]
    float* share_l = colsum_l + 4 * 64;                // [4] the waves' shares of sum_i cw_i * rowloss_i

    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const int swz = (c >> 1) & 7;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(c * BKB) + (uint32_t)(((kk * 2 + g) ^ swz) << 4);
    const uint32_t a_row_off = (uint32_t)((wave & 1) * 32 * BKB);        // this wave's 32 rows inside its A half slot
    uint32_t cs_off = 0;                               // ring byte offset of the next slot to consume

    i32x4 fa[DecAst::MAXKT][4];                        // this wave's h fragments: rows [32 wave, +32) of the panel, all of K
#pragma unroll
    for (int kt = 0; kt < DecAst::MAXKT; ++kt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fa[kt][kk] = i32x4{0, 0, 0, 0};
    int tm_loaded = -1;
    for (int it = 0; it < ntiles; ++it) {
        const int ti = i0 + it;
        const int tm = ti / nc, tn = c0 + ti % nc;
        // ---- tile prologue: what the epilogue needs from memory is requested now and parked in a few registers across the K loop ----
        uint32_t xb = 0;
        if constexpr (XBITS) xb = e.x_bits[(int64_t)(tm * BM + wave * 32 + (lane & 31)) * e.ldxb + tn * 2 + (lane >> 5)];
        const float cw_r = e.cw[tm * BM + wave * 32 + (lane & 31)];                 // zero beyond B by construction
        float bvv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bvv[nt] = (tn * BN_T + c + nt * 32) < e.F ? e.bv[tn * BN_T + c + nt * 32] : 0.f;

        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

        // ---- the tile's runs, one ring slot per step: passes 0 / 1 move the new panel's h fragments ring -> registers (the waves of that row half read),
        //      passes 2 + s multiply K segment s.  One barrier per step; the body of K tile kt is selected by a uniform switch so that fa[kt] stays a
        //      compile-time register index while the loop itself is NOT unrolled ----
        auto ring_step = [&]() -> uint32_t {
            // the slot to consume has landed once at most the pieces of the L - 1 younger slots are outstanding (exactly L are issued ahead while the
            // stream lasts; behind its end everything outstanding is waited for)
            if (p_more) wait_vm<2 * (L - 1)>(); else wait_vm<0>();
            __builtin_amdgcn_s_barrier();              // it landed for every wave, and every wave is done reading the slot consumed one step ago
            asm volatile("" ::: "memory");
            if (p_more) dma_one();                     // refill that slot: L stay in flight
            const uint32_t sb = lbase + cs_off;
            cs_off += SLOT;
            if (cs_off == DecAst::RING) cs_off = 0;
            return sb;
        };
        if (tm != tm_loaded) {                         // a new panel: runs 0 / 1 move its h fragments ring -> registers (the waves of that row half read)
#pragma nounroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int kt = 0; kt < DecAst::MAXKT; ++kt) {
                    if (kt < nkt) {
                        const uint32_t sb = ring_step();
                        if ((wave >> 1) == half && !(e.dbg & 1)) {
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) fa[kt][kk] = lds_read_b128(sb + a_row_off + so[kk]);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                    }
                }
            }
            stamp();
        }
        // runs 2 + s multiply K segment s, one ring slot per step.  The body of K tile kt is selected by a uniform switch so that fa[kt] stays a
        // compile-time register index while the loop itself is NOT unrolled (one copy of the ring step)
        for (int sg = 0; sg < nseg; ++sg) {
#pragma nounroll
            for (int kt = 0; kt < nkt; ++kt) {
                const uint32_t sb = ring_step();
                if (!(e.dbg & 2)) {
                    // B fragments of k-step kk for both column blocks; two k-steps are in flight at a time (16 registers, not 32: beside the 128 of h
                    // the full set pushed one K tile of h into scratch)
                    i32x4 f0a = lds_read_b128(sb + so[0]), f0b = lds_read_b128_off4096(sb + so[0]);
                    i32x4 f1a = lds_read_b128(sb + so[1]), f1b = lds_read_b128_off4096(sb + so[1]);
                    i32x4 f2a, f2b, f3a, f3b;
#define DAE_AST_MM(KT, KK, FA, FB)                               \
    __builtin_amdgcn_sched_barrier(0);                           \
    Mma<T>::run(fa[KT][KK], FA, acc[0]);                         \
    Mma<T>::run(fa[KT][KK], FB, acc[1]);                         \
    __builtin_amdgcn_sched_barrier(0);
#define DAE_AST_BCASE(KT)                                                                                       \
    case KT:                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                                                      \
        DAE_AST_MM(KT, 0, f0a, f0b)                                                                             \
        f2a = lds_read_b128(sb + so[2]); f2b = lds_read_b128_off4096(sb + so[2]);                               \
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                                                      \
        DAE_AST_MM(KT, 1, f1a, f1b)                                                                             \
        f3a = lds_read_b128(sb + so[3]); f3b = lds_read_b128_off4096(sb + so[3]);                               \
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                                                      \
        DAE_AST_MM(KT, 2, f2a, f2b)                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        DAE_AST_MM(KT, 3, f3a, f3b)                                                                             \
        break;
                    switch (kt) { DAE_AST_BCASE(0) DAE_AST_BCASE(1) DAE_AST_BCASE(2) DAE_AST_BCASE(3) DAE_AST_BCASE(4) DAE_AST_BCASE(5) DAE_AST_BCASE(6) DAE_AST_BCASE(7) default: break; }
#undef DAE_AST_BCASE
#undef DAE_AST_MM
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the default case: no read may stay pending into the next step)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        tm_loaded = tm;

        stamp();
        // ---- epilogue of tile (tm, tn): gemm_decode_loss's arithmetic on the wave layout 4 x 1 (wave w: rows [32 w, +32), all 64 columns), wave-local ----
        // Everything below is recomputed per tile from a laundered lane id: hoisted out of the tile loop, the epilogue's per-lane addresses and constants
        // (~60 VGPRs) sat beside the 128 A registers through the K loop and went to scratch -- and a scratch reload in a kernel with two waves per SIMD is
        // a ~1 us stall each (measured: 22 us of the kernel).
        {
        int lane_l = threadIdx.x & 63;
        asm volatile("" : "+v"(lane_l));
        const int lane = lane_l, g = lane >> 5, c = lane & 31;
        const int row0 = tm * BM + wave * 32;          // first batch row of this wave
        if (lane < 32) cw_l[lane] = cw_r;
        if constexpr (XBITS) {
            xb_l[(lane & 31) * 2 + (lane >> 5)] = xb;
        } else {                                       // valued clean rows: this wave's [32][64] tile, fetched now (binary input is the hot case)
            i32x4 xr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = lane + 64 * i;
                xr[i] = *reinterpret_cast<const i32x4*>(X + (int64_t)(row0 + (ch >> 3)) * e.ldx + tn * BN_T + (ch & 7) * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = lane + 64 * i;
                *reinterpret_cast<i32x4*>(R0 + (ch >> 3) * P0 + (ch & 7) * 16) = xr[i];
            }
        }
        if constexpr (IS_COS) {
            if (lane < 32) {
                inx_l[lane] = rsqrtf(fmaxf(e.cos_stats[row0 + lane], 1e-12f));
                cyy_l[lane] = e.cos_pass == 2 ? e.cos_stats[e.Bp + row0 + lane] : 0.f;
                cxy_l[lane] = e.cos_pass == 2 ? e.cos_stats[2 * e.Bp + row0 + lane] : 0.f;
            }
        }
        const int lrow0 = 4 * g;                       // row of r = 0 inside the wave's 32
        const float eps = 1e-16f;
        float colsum[NT], cm[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            colsum[nt] = 0.f;
            cm[nt] = (tn * BN_T + c + nt * 32) < e.F ? 1.f : 0.f;
        }
        char* r0_lane = R0 + lrow0 * P0 + c * 2;
        T* d2t_lane = D2T ? D2T + (int64_t)(tn * BN_T + c) * e.lddt + row0 + lrow0 : nullptr;
        const bool want_rows = e.rowloss_part != nullptr;
        const float osc = e.op_scale;
        float wl_acc = 0.f;                            // this lane's share of sum_i cw_i * loss_if
        auto epi_block = [&](auto R4, auto FASTV) {
            constexpr int r4 = decltype(R4)::value;
            constexpr bool FAST = decltype(FASTV)::value;
            constexpr int rloc = 8 * r4;
            float d2v[NT][4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int r = r4 * 4 + qq;
                const int lrow = lrow0 + rloc + qq;
                const float cwi = cw_l[lrow];
                float rl = 0.f, s_yy = 0.f, s_xy = 0.f;
                float inx = 0.f, cs_yy = 0.f, cs_xy = 0.f;
                if constexpr (IS_COS) { inx = inx_l[lrow]; cs_yy = cyy_l[lrow]; cs_xy = cxy_l[lrow]; }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float z = acc[nt][r] + bvv[nt];
                    float xv;
                    if constexpr (XBITS) xv = (float)((xb_l[lrow * 2 + nt] >> c) & 1u);
                    else xv = bf2f(*reinterpret_cast<const bf16_t*>(r0_lane + (rloc + qq) * P0 + nt * 64));
                    float l = 0.f, dy = 0.f;
                    if constexpr (FAST) {
                        const float en = __builtin_amdgcn_exp2f(-fabsf(z) * kLog2e);          // exp(-|z|)
                        const float op = 1.0f + en;
                        const float rr = __builtin_amdgcn_rcpf(op);
                        const float yv = z >= 0.f ? rr : en * rr;
                        l = kLn2 * __builtin_amdgcn_logf(op) + fmaxf(z, 0.f) - xv * z;
                        const float d2 = (cwi * cm[nt]) * (yv - xv);
                        rl += cm[nt] * l;
                        colsum[nt] += d2;
                        const float d2s = sat16(d2 * osc);                  // the 16-bit images hold op_scale * delta2
                        d2v[nt][qq] = d2s;
                        *reinterpret_cast<bf16_t*>(r0_lane + (rloc + qq) * P0 + nt * 64) = f2bf_hw(d2s);
                        continue;
                    }
                    const float y = act_fwd<ACT>(z);
                    if constexpr (LOSS == DAE_LOSS_CROSS_ENTROPY) {
                        const float a = y + eps, bb = (1.0f - y) + eps;         // reference op order: (1.-y)+1e-16
                        const float la = __builtin_amdgcn_logf(a), lb = __builtin_amdgcn_logf(bb);
                        l = -kLn2 * (xv * la + (1.0f - xv) * lb);
                        dy = (1.0f - xv) * __builtin_amdgcn_rcpf(bb) - xv * __builtin_amdgcn_rcpf(a);
                    } else if constexpr (LOSS == DAE_LOSS_MEAN_SQUARED) {
                        const float d = xv - y;
                        l = d * d;
                        dy = -2.0f * d;
                    } else {
                        const float xh = xv * inx;
                        if (pass1) {
                            s_yy += cm[nt] * y * y;
                            s_xy += cm[nt] * xh * y;
                        } else {
                            const float big = cs_yy >= 1e-12f ? 1.f : 0.f;      // tf.maximum routes grad to sum y^2 iff >= eps
                            const float s = rsqrtf(fmaxf(cs_yy, 1e-12f));
                            dy = -(xh * s - big * cs_xy * s * s * s * y);
                        }
                    }
                    const float d2 = pass1 ? 0.f : (cwi * cm[nt]) * dy * act_bwd<ACT>(y);   // cw is 0 on padded rows
                    rl += cm[nt] * l;
                    colsum[nt] += d2;
                    const float d2s = sat16(d2 * osc);
                    d2v[nt][qq] = d2s;
                    *reinterpret_cast<bf16_t*>(r0_lane + (rloc + qq) * P0 + nt * 64) = f2bf_hw(d2s);
                }
                // wavefront (DPP) sum over the 32 lanes that share this row; lanes 16..31 / 48..63 hold it
                if constexpr (IS_COS) {
                    if (pass1) {
                        s_yy = half32_sum_hi(s_yy); s_xy = half32_sum_hi(s_xy);
                        if (c == 31) { rs_l[lrow] = s_yy; xy_l[lrow] = s_xy; }
                    }
                } else {
                    wl_acc += cwi * rl;
                    if (want_rows) {
                        rl = half32_sum_hi(rl);
                        if (c == 31) rs_l[lrow] = rl;
                    }
                }
            }
            // delta2^T straight from the accumulator layout: four consecutive batch rows of feature (c + 32 nt) are 8 contiguous bytes of its image row
            // (the g = 0 / g = 1 lanes of a column write adjacent pieces; a column's 64 bytes of this wave complete within the four blocks)
            if (d2t_lane && !pass1 && !(e.dbg & 32)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    uint2 v;
                    v.x = f2bf_pack_hw(d2v[nt][0], d2v[nt][1]);
                    v.y = f2bf_pack_hw(d2v[nt][2], d2v[nt][3]);
                    *reinterpret_cast<uint2*>(d2t_lane + (int64_t)nt * 32 * e.lddt + rloc) = v;
                }
            }
            // one block's sums are closed before the next block starts: left free, the scheduler sank all 16 column-sum / loss-share additions behind the
            // four blocks and carried their 48 operands there -- through scratch, with 128 registers of h resident
            asm volatile("" : "+v"(colsum[0]), "+v"(colsum[1]), "+v"(wl_acc));
            __builtin_amdgcn_sched_barrier(0);
        };
        // this wave's 32 rows are pure batch padding (rows >= B): delta2 = 0 without evaluating the loss (cw is 0 there)
        const int vblk = e.no_pad_skip ? 4 : min(4, (e.B - tm * BM + 31) >> 5);
        bool fast = false;
        if constexpr (LOSS == DAE_LOSS_CROSS_ENTROPY && ACT == DAE_ACT_SIGMOID) {
            float zmax = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) zmax = fmaxf(zmax, fabsf(acc[nt][r] + bvv[nt]));
            fast = __builtin_amdgcn_ballot_w64(!(zmax < CE_FAST_ZMAX)) == 0ull && !e.ce_literal;   // NaN logits take the literal path
        }
        if (wave >= vblk || (e.dbg & 8)) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<bf16_t*>(r0_lane + (8 * r4 + qq) * P0 + nt * 64) = (bf16_t)0;
                    uint2 z; z.x = 0u; z.y = 0u;
                    if (d2t_lane && !pass1 && !(e.dbg & 32)) *reinterpret_cast<uint2*>(d2t_lane + (int64_t)nt * 32 * e.lddt + 8 * r4) = z;
                }
            if (lane < 32) { rs_l[lane] = 0.f; xy_l[lane] = 0.f; }
        } else if (fast) {
            if constexpr (LOSS == DAE_LOSS_CROSS_ENTROPY && ACT == DAE_ACT_SIGMOID) {
                epi_block(std::integral_constant<int, 0>{}, std::true_type{});
                epi_block(std::integral_constant<int, 1>{}, std::true_type{});
                epi_block(std::integral_constant<int, 2>{}, std::true_type{});
                epi_block(std::integral_constant<int, 3>{}, std::true_type{});
            }
        } else {
            epi_block(std::integral_constant<int, 0>{}, std::false_type{});
            epi_block(std::integral_constant<int, 1>{}, std::false_type{});
            epi_block(std::integral_constant<int, 2>{}, std::false_type{});
            epi_block(std::integral_constant<int, 3>{}, std::false_type{});
        }
        if (!pass1) {                                       // column sums of this wave's 32 rows: rows of g = 0 and g = 1, then one lane per column
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float v = colsum[nt] + __shfl_xor(colsum[nt], 32, 64);
                if (g == 0) colsum_l[wave * BN_T + c + nt * 32] = v;
            }
        }
        if constexpr (!IS_COS) {
            if (e.tile_part) {
                const float v = wave64_sum_hi(wl_acc);
                if (lane == 63) share_l[wave] = v;
            }
        }
        // ---- this wave's rows leave the CU: delta2 as coalesced 16-byte pieces of 128-byte row runs, the per-row partial sums ----
        if (!pass1 && D2 && !(e.dbg & 16)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = lane + 64 * i;
                *reinterpret_cast<i32x4*>(D2 + (int64_t)(row0 + (ch >> 3)) * e.ldd + tn * BN_T + (ch & 7) * 8) =
                    *reinterpret_cast<const i32x4*>(R0 + (ch >> 3) * P0 + (ch & 7) * 16);
            }
        }
        {
            // two partial rows per column tile in the consumers' layout: [0] = the row sums of the 64 columns, [1] = 0
            const int k = lane & 31, w = lane >> 5;
            const bool rowok = (row0 + k) < e.B;
            if constexpr (IS_COS) {
                if (pass1) {
                    e.cos_part[(int64_t)(tn * 2 + w) * e.Bp + row0 + k] = (rowok && w == 0) ? rs_l[k] : 0.f;
                    e.cos_part[(int64_t)(2 * p.tiles_n + tn * 2 + w) * e.Bp + row0 + k] = (rowok && w == 0) ? xy_l[k] : 0.f;
                }
            } else {
                if (e.rowloss_part) e.rowloss_part[(int64_t)(tn * 2 + w) * e.Bp + row0 + k] = (rowok && w == 0) ? rs_l[k] : 0.f;
            }
        }
        lds_barrier();                                      // the four waves' column sums and loss shares are in LDS
        if (e.dbv_part && !pass1 && wave < 2 && !(e.dbg & 64)) {             // two partial rows per row tile: rows [0, 64) and [64, 128) of the panel; wave w2 adds its pair
            e.dbv_part[(int64_t)(tm * 2 + wave) * e.Fp + tn * BN_T + lane] = colsum_l[(2 * wave) * BN_T + lane] + colsum_l[(2 * wave + 1) * BN_T + lane];
        }
        if constexpr (!IS_COS) {
            if (e.tile_part && wave == 2 && lane == 0) e.tile_part[tm * p.tiles_n + tn] = (share_l[0] + share_l[1]) + (share_l[2] + share_l[3]);
        }
        stamp();
        // (the next tile's first ring_step barrier orders these LDS reads before the next epilogue's writes)
        }   // (laundered-id scope of the epilogue)
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static int g_nst = 2;   // staging variant of the plain GEMM: 0 register-staged, 2/3/4 global_load_lds ring depth

// one K segment of a contraction as the host hands it over: A_seg [M x K], Bt_seg [N x K], both K-contiguous, leading dimensions in elements
static int fill_params_n(GemmParams& p, int dtype, int M, int N, const GemmSegDesc* segs, int nsegs, int splits, int bn = BN, bool a_is_k_by_m = false) {
    // a_is_k_by_m (gemm_dw_pc<TRA>): A_seg is a row-major [K x M] image -- its 32-bit DMA offsets span one K tile (64 rows), not M rows
    const int es = (dtype == DAE_BF16) ? 2 : 4;
    const int kel = BKB / es;
    DAE_CHECK_ARG(dtype == DAE_BF16 || dtype == DAE_F32, "gemm: bad dtype %d", dtype);
    DAE_CHECK_ARG(M > 0 && N > 0 && M % BM == 0 && N % bn == 0, "gemm: M=%d N=%d must be positive multiples of %d / %d", M, N, BM, bn);
    DAE_CHECK_ARG(segs && nsegs >= 1 && nsegs <= GEMM_MAX_SEG, "gemm: %d K segments (1..%d)", nsegs, GEMM_MAX_SEG);
    DAE_CHECK_ARG(segs[0].K > 0, "gemm: the first K segment is empty");
    memset(p.seg, 0, sizeof(p.seg));
    memset(p.bt2, 0, sizeof(p.bt2));
    p.nseg = 0; p.ktiles_total = 0; p.epi_vec = 0; p.out_scale = 1.f;
    for (int i = 0; i < nsegs; ++i) {
        const GemmSegDesc& d = segs[i];
        DAE_CHECK_ARG(d.K >= 0 && d.K % kel == 0, "gemm: K of segment %d = %d must be a multiple of %d", i, d.K, kel);
        if (d.K == 0) continue;                                   // empty segments are dropped (seg[] holds the non-empty ones in order)
        DAE_CHECK_ARG(d.A && d.Bt, "gemm: null operand in segment %d", i);
        DAE_CHECK_ARG((d.lda * es) % 16 == 0 && (d.ldb * es) % 16 == 0, "gemm: leading dimensions must be 16-byte multiples");
        DAE_CHECK_ARG(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.Bt % 16) == 0, "gemm: operands must be 16-byte aligned");
        DAE_CHECK_ARG((uint64_t)(a_is_k_by_m ? 64 : M) * (uint64_t)(d.lda * es) < (1ull << 32) && (uint64_t)N * (uint64_t)(d.ldb * es) < (1ull << 32),
                      "gemm: an operand panel (rows x leading dimension) must stay below 4 GiB (32-bit DMA offsets)");
        p.seg[p.nseg++] = {(const char*)d.A, (const char*)d.Bt, d.lda * es, d.ldb * es, d.K / kel};
        p.ktiles_total += d.K / kel;
    }
    p.tiles_m = M / BM; p.tiles_n = N / bn;
    p.splits = splits < 1 ? 1 : splits;
    p.trace = nullptr;
    DAE_CHECK_ARG(p.splits <= p.ktiles_total, "gemm: splits=%d exceeds k-tiles=%d", p.splits, p.ktiles_total);
    return 0;
}
static int fill_params(GemmParams& p, int dtype, int M, int N, const void* A0, int64_t lda0, const void* Bt0,
                       int64_t ldb0, int K0, const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int K1,
                       int splits, int bn = BN) {
    const GemmSegDesc segs[2] = {{A0, lda0, Bt0, ldb0, K0}, {A1, lda1, Bt1, ldb1, K1}};
    return fill_params_n(p, dtype, M, N, segs, 2, splits, bn);
}

// bf16 decode runs on 128 x 64 tiles (three workgroups per CU), fp32 (parity mode) keeps the 128 x 128 tile
constexpr int DECODE_BN_BF16 = 64;
int decode_tile_n(int dtype) { return dtype == DAE_BF16 ? DECODE_BN_BF16 : BN; }

typedef void (*f32out_fn)(GemmParams, float*, int64_t, int64_t);
constexpr int DEFAULT_NST = 2;
template <typename T> static f32out_fn f32out_kernel(int nst, int role) {
    if (nst == DEFAULT_NST) {
        switch (role) {
            case ROLE_ENCODE: return gemm_nt_f32out<T, DEFAULT_NST, ROLE_ENCODE>;
            case ROLE_DH: return gemm_nt_f32out<T, DEFAULT_NST, ROLE_DH>;
            case ROLE_DW: return gemm_nt_f32out<T, DEFAULT_NST, ROLE_DW>;
            case ROLE_GRAM: return gemm_nt_f32out<T, DEFAULT_NST, ROLE_GRAM>;
            default: break;
        }
    }
    switch (nst) {
        case 0: return gemm_nt_f32out<T, 0, ROLE_GENERIC>;
        case 3: return gemm_nt_f32out<T, 3, ROLE_GENERIC>;
        case 4: return gemm_nt_f32out<T, 4, ROLE_GENERIC>;
        default: return gemm_nt_f32out<T, 2, ROLE_GENERIC>;
    }
}
typedef void (*pc_fn)(GemmParams, float*, int64_t, int64_t, LabelJob, int);
template <typename T> static pc_fn pc_kernel(int role) {
    switch (role) {
        case ROLE_ENCODE: return gemm_nt_pc<T, PC_NST, ROLE_ENCODE>;
        case ROLE_DH: return gemm_nt_pc<T, PC_NST, ROLE_DH>;
        case ROLE_DW: return gemm_nt_pc<T, PC_NST, ROLE_DW>;
        case ROLE_GRAM: return gemm_nt_pc<T, PC_NST, ROLE_GRAM>;
        default: return gemm_nt_pc<T, PC_NST, ROLE_GENERIC>;
    }
}
static int g_cus = 0;        // compute units of the current device (set by gemm_init)
static int g_dw_pc = 1;      // dW + optimizer on the 160 x 128 producer/consumer kernel when its grid fills one round (dae_set_glds(-3/-4))
static int g_dw_rounds = 1;  // ... or at most this many rounds of the chip (dae_set_glds(-100 - r)): one workgroup per CU at a time, the next tile's K loop starts when a CU
                             // frees -- the optimizer stays in the epilogue for shapes beyond 256 tiles (F = 50000: 2504 tiles) instead of a gradient round trip
                             // through HBM + a separate optimizer launch
static int g_dw_rounds_split = 16;   // the split 16-bit modes take the multi-round form by default: measured at 896 x 50048 x 1024 (c4, f16x2) 357 us fused against
                                     // 246 us GEMM to memory + 208 us optimizer kernel (profiles/r05_c4_dw_rounds.txt)
static int g_use_pc = 1;     // dae_set_glds(-1) keeps the 4-wave kernel for every grid (A/B)
static int g_pad_skip = 1;   // skip the MFMAs / loss evaluation of 32-row blocks that are pure batch padding (dae_set_glds(-11) off, (-12) on: A/B; plan option "pad_skip")
static int g_pc_vec = 1;     // gemm_nt_pc epilogue: 1 = LDS-staged 16-byte pieces (default), 0 = dword stores (dae_set_glds(-9) / (-10))
typedef void (*decode_fn)(GemmParams, DecodeEpi);
static decode_fn decode_kernel_xbits(int loss, int act) {
#define DAE_DKX(LV, AV) if (loss == LV && act == AV) return gemm_decode_loss<bf16_t, LV, AV, true, DECODE_BN_BF16>;
    DAE_DKX(0, 0) DAE_DKX(0, 1) DAE_DKX(0, 2) DAE_DKX(1, 0) DAE_DKX(1, 1) DAE_DKX(1, 2) DAE_DKX(2, 0) DAE_DKX(2, 1) DAE_DKX(2, 2)
#undef DAE_DKX
    return nullptr;
}
// split-bf16 mode: the same kernels with the lo images of delta2 / delta2^T as extra outputs (bf16, 64-column tiles)
static decode_fn decode_kernel_res(int loss, int act, bool xbits) {
#define DAE_DKR(LV, AV)                                                                                       \
    if (loss == LV && act == AV)                                                                              \
        return xbits ? gemm_decode_loss<bf16_t, LV, AV, true, DECODE_BN_BF16, true> : gemm_decode_loss<bf16_t, LV, AV, false, DECODE_BN_BF16, true>;
    DAE_DKR(0, 0) DAE_DKR(0, 1) DAE_DKR(0, 2) DAE_DKR(1, 0) DAE_DKR(1, 1) DAE_DKR(1, 2) DAE_DKR(2, 0) DAE_DKR(2, 1) DAE_DKR(2, 2)
#undef DAE_DKR
    return nullptr;
}
// 128-column tiles for the 16-bit modes (DecodeEpi::bn = 128): half the LDS-DMA instructions per MFMA of the 64-column tile (A 16 KiB + B 16 KiB per 16 MFMAs
// of a wave instead of A 16 KiB + B 8 KiB per 8) at two workgroups per CU -- pays when the tile count is several rounds of the chip (F = 50000) or the K
// loop is long (the split modes' extra product terms); the lo images of delta2 (RES) exist for the 64-column form only
static decode_fn decode_kernel_wide(int loss, int act, bool xbits) {
#define DAE_DKW(LV, AV)                                                                                       \
    if (loss == LV && act == AV)                                                                              \
        return xbits ? gemm_decode_loss<bf16_t, LV, AV, true, BN> : gemm_decode_loss<bf16_t, LV, AV, false, BN>;
    DAE_DKW(0, 0) DAE_DKW(0, 1) DAE_DKW(0, 2) DAE_DKW(1, 0) DAE_DKW(1, 1) DAE_DKW(1, 2) DAE_DKW(2, 0) DAE_DKW(2, 1) DAE_DKW(2, 2)
#undef DAE_DKW
    return nullptr;
}
// the 64-column kernel with the paired K loop (two K segments that share their A operand: the split modes' h.W_hi + h.W_lo)
static decode_fn decode_kernel_pair(int loss, int act, bool xbits) {
#define DAE_DKP(LV, AV)                                                                                       \
    if (loss == LV && act == AV)                                                                              \
        return xbits ? gemm_decode_loss<bf16_t, LV, AV, true, DECODE_BN_BF16, false, true> : gemm_decode_loss<bf16_t, LV, AV, false, DECODE_BN_BF16, false, true>;
    DAE_DKP(0, 0) DAE_DKP(0, 1) DAE_DKP(0, 2) DAE_DKP(1, 0) DAE_DKP(1, 1) DAE_DKP(1, 2) DAE_DKP(2, 0) DAE_DKP(2, 1) DAE_DKP(2, 2)
#undef DAE_DKP
    return nullptr;
}
// the 64-column kernel with the three-term K loop (mainloop_n64_x3: (h_hi, W_hi) (h_hi, W_lo) (h_lo, W_hi) in two stages per K tile)
// (binary input = the bit image of x only: with the 16-byte x prefetch registers on top the loop does not fit the 168 VGPRs of three workgroups per CU)
static decode_fn decode_kernel_x3(int loss, int act) {
#define DAE_DK3(LV, AV) if (loss == LV && act == AV) return gemm_decode_loss<bf16_t, LV, AV, true, DECODE_BN_BF16, false, false, true>;
    DAE_DK3(0, 0) DAE_DK3(0, 1) DAE_DK3(0, 2) DAE_DK3(1, 0) DAE_DK3(1, 1) DAE_DK3(1, 2) DAE_DK3(2, 0) DAE_DK3(2, 1) DAE_DK3(2, 2)
#undef DAE_DK3
    return nullptr;
}
// ... and with the two-term register-carry loop (mainloop_n64_c2: (h, W_hi) (h, W_lo)); binary input, with or without the lo images of delta2
static decode_fn decode_kernel_c2(int loss, int act, bool res) {
#define DAE_DKC(LV, AV)                                                                                       \
    if (loss == LV && act == AV)                                                                              \
        return res ? gemm_decode_loss<bf16_t, LV, AV, true, DECODE_BN_BF16, true, false, false, true> : gemm_decode_loss<bf16_t, LV, AV, true, DECODE_BN_BF16, false, false, false, true>;
    DAE_DKC(0, 0) DAE_DKC(0, 1) DAE_DKC(0, 2) DAE_DKC(1, 0) DAE_DKC(1, 1) DAE_DKC(1, 2) DAE_DKC(2, 0) DAE_DKC(2, 1) DAE_DKC(2, 2)
#undef DAE_DKC
    return nullptr;
}
// the A-stationary persistent kernel (gemm_decode_ast)
static decode_fn decode_kernel_ast(int loss, int act, bool xbits) {
#define DAE_DKA(LV, AV)                                                                                       \
    if (loss == LV && act == AV)                                                                              \
        return xbits ? gemm_decode_ast<LV, AV, true> : gemm_decode_ast<LV, AV, false>;
    DAE_DKA(0, 0) DAE_DKA(0, 1) DAE_DKA(0, 2) DAE_DKA(1, 0) DAE_DKA(1, 1) DAE_DKA(1, 2) DAE_DKA(2, 0) DAE_DKA(2, 1) DAE_DKA(2, 2)
#undef DAE_DKA
    return nullptr;
}
static int g_decode_dbg = 0;       // dae_set_glds(-500000 - bits): timing probes of gemm_decode_ast (DecodeEpi::dbg)
static int g_decode_x3 = 1;        // dae_set_glds(-17) off / (-18) on; plan option "decode_x3": the three-term decode (f16x2h) on mainloop_n64_x3 instead of three K segments
static int g_decode_ast = 0;       // dae_set_glds(-15) off / (-16) on; plan option "decode_ast".  OFF: measured 44 us against 38 for the tile kernel at c2 (profiles/r06_decode_ast.txt)
constexpr int DECODE_PAIR_LDS = 2 * (TILE_BYTES + 2 * 64 * BKB) > DecGeo<DECODE_BN_BF16>::EPI_BYTES ? 2 * (TILE_BYTES + 2 * 64 * BKB) : DecGeo<DECODE_BN_BF16>::EPI_BYTES;
static int g_decode_pair = 0;      // dae_set_glds(-13) off / (-14) on; plan option "decode_pair"
template <typename T> static decode_fn decode_kernel(int loss, int act) {
#define DAE_DK(LV, AV) if (loss == LV && act == AV) return gemm_decode_loss<T, LV, AV, false, (sizeof(T) == 2 ? DECODE_BN_BF16 : BN)>;
    DAE_DK(0, 0) DAE_DK(0, 1) DAE_DK(0, 2) DAE_DK(1, 0) DAE_DK(1, 1) DAE_DK(1, 2) DAE_DK(2, 0) DAE_DK(2, 1) DAE_DK(2, 2)
#undef DAE_DK
    return nullptr;
}
static int gemm_init() {
    static int rc = [] {
        const int nsts[4] = {0, 2, 3, 4};
        for (int n : nsts)
            for (int role = 0; role <= ROLE_GRAM; ++role) {
                DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f32out_kernel<bf16_t>(n, role)),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_for(n)));
                DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f32out_kernel<float>(n, role)),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_for(n)));
            }
        for (int role = 0; role <= ROLE_GRAM; ++role) {
            DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pc_kernel<bf16_t>(role)),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_for(PC_NST)));
            DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pc_kernel<float>(role)),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_for(PC_NST)));
        }
        {
            int dev = 0;
            DAE_CHECK_HIP(hipGetDevice(&dev));
            DAE_CHECK_HIP(hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev));
        }
        for (int l = 0; l < 3; ++l)
            for (int a = 0; a < 3; ++a) {
                DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel<bf16_t>(l, a)),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, DecGeo<DECODE_BN_BF16>::LDS_BYTES));
                DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel<float>(l, a)),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, DecGeo<BN>::LDS_BYTES));
                DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel_xbits(l, a)),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, DecGeo<DECODE_BN_BF16>::LDS_BYTES));
            }
        return 0;
    }();
    return rc;
}

static int g_w8 = 1;         // 256 x 256 / 8-MFMA-wave kernel for the large split-K contractions (dae_set_glds(-6) disables, -7 enables)
typedef void (*w8_fn)(W8Params, float*, int64_t, int64_t);
static w8_fn w8_kernel(int role) {
    switch (role) {
        case ROLE_ENCODE: return gemm_nt_w8<ROLE_ENCODE>;
        case ROLE_DH: return gemm_nt_w8<ROLE_DH>;
        case ROLE_DW: return gemm_nt_w8<ROLE_DW>;
        case ROLE_GRAM: return gemm_nt_w8<ROLE_GRAM>;
        default: return gemm_nt_w8<ROLE_GENERIC>;
    }
}
// Does the 256 x 256 kernel pay for this shape, and with how many K slices?  bf16, at most one workgroup per CU and at least 60 % of
// the CUs busy, >= 16 K tiles per workgroup (its 2-tile prologue and the 256 KiB slab it writes must be amortised); slices in
// multiples of 8 so that one slice maps to one XCD.  0 = the shape stays on the 128 x 128 kernels.  The plan sizes its slab
// workspace with this number, and launch_gemm_f32out takes the 256 x 256 path exactly when it is handed the same number.
int gemm_w8_splits(int dtype, int M, int N, int ktiles) {
    if (gemm_init()) return 0;
    if (!g_w8 || dtype != DAE_BF16 || g_cus <= 0) return 0;
    const int tiles = ((M + W8_BM - 1) / W8_BM) * ((N + W8_BN - 1) / W8_BN);
    if (M < 512 || N < 512 || tiles > g_cus) return 0;
    int s = g_cus / tiles;
    if (s >= 8) s = (s / 8) * 8;
    if (s > 16) s = 16;
    while (s > 1 && ktiles / s < 16) s = s > 8 ? s - 8 : s / 2;
    if (s < 1 || ktiles / s < 16 || 10 * tiles * s < 6 * g_cus) return 0;
    return s;
}
int launch_gemm_f32out(int dtype, int M, int N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0, int K0,
                       const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int K1, float* C, int64_t ldc,
                       int splits, int64_t slab_stride, hipStream_t st, int role, const LabelJob* label_job, int* label_done, int m_valid) {
    const GemmSegDesc segs[2] = {{A0, lda0, Bt0, ldb0, K0}, {A1, lda1, Bt1, ldb1, K1}};
    return launch_gemm_f32out_n(dtype, M, N, segs, 2, C, ldc, splits, slab_stride, st, role, label_job, label_done, 1.f, m_valid);
}

int launch_gemm_f32out_n(int dtype, int M, int N, const GemmSegDesc* segs, int nsegs, float* C, int64_t ldc, int splits, int64_t slab_stride,
                         hipStream_t st, int role, const LabelJob* label_job, int* label_done, float out_scale, int m_valid) {
    if (label_done) *label_done = 0;
    GemmParams p;
    if (int rc = fill_params_n(p, dtype, M, N, segs, nsegs, splits)) return rc;
    DAE_CHECK_ARG(C != nullptr, "gemm: C is null");
    DAE_CHECK_ARG(out_scale == 1.f || p.splits == 1, "gemm: out_scale applies to un-split launches (slabs are scaled by the kernel that sums them)");
    p.out_scale = out_scale;
    (void)m_valid;          // (rows of A that hold data: reserved -- skipping the all-padding MFMA blocks was measured worth < 1 % and is not done, see gemm_decode_loss)
    if (int rc = gemm_init()) return rc;
    if (out_scale == 1.f && p.splits == gemm_w8_splits(dtype, M, N, p.ktiles_total)) {      // (the 256 x 256 kernel has no output scale)
        W8Params q;
        for (int i = 0; i < GEMM_MAX_SEG; ++i) q.seg[i] = p.seg[i];
        q.nseg = p.nseg; q.ktiles_total = p.ktiles_total; q.M = M; q.N = N;
        q.tiles_m = (M + W8_BM - 1) / W8_BM; q.tiles_n = (N + W8_BN - 1) / W8_BN; q.splits = p.splits;
        const int ws = p.splits;
        static const int w8_attr_rc = [] {
            int rc = 0;
            for (int r = 0; r <= ROLE_GRAM; ++r)
                rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(w8_kernel(r)), hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
            return rc;
        }();
        DAE_CHECK_ARG(w8_attr_rc == 0, "gemm: hipFuncSetAttribute failed for the 256 x 256 kernel (%d)", w8_attr_rc);
        DAE_LAUNCH(w8_kernel(role), dim3(q.tiles_m * q.tiles_n * ws), dim3(W8_THREADS), W8_LDS, st, q, C, ldc, slab_stride);
        DAE_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid(grid_blocks(p)), block(GEMM_THREADS);
    const int nst = g_nst;
    // (its epilogue writes 16-byte pieces: C, the leading dimension and the slab stride must keep them aligned -- else the 4-wave kernel's dword stores)
    const bool c_vec_ok = ((uintptr_t)C % 16) == 0 && ldc % 4 == 0 && slab_stride % 4 == 0;
    if (nst == DEFAULT_NST && (int)grid.x <= g_cus && g_use_pc) {      // at most one workgroup per CU: producer/consumer waves
        pc_fn kp = dtype == DAE_BF16 ? pc_kernel<bf16_t>(role) : pc_kernel<float>(role);
        p.epi_vec = (g_pc_vec && c_vec_ok) ? 1 : 0;
        LabelJob job; memset(&job, 0, sizeof(job));
        const bool with_labels = label_job && label_job->Bp <= 1024 && (int)grid.x < g_cus;      // a CU must be free for it
        if (with_labels) job = *label_job;
        DAE_LAUNCH(kp, dim3(grid.x + (with_labels ? 1 : 0)), dim3(PC_THREADS), lds_bytes_for(PC_NST), st, p, C, ldc, slab_stride, job,
                           with_labels ? (int)grid.x : -1);
        DAE_CHECK_LAUNCH();
        if (with_labels && label_done) *label_done = 1;
        return 0;
    }
    f32out_fn k = dtype == DAE_BF16 ? f32out_kernel<bf16_t>(nst, role) : f32out_kernel<float>(nst, role);
    DAE_LAUNCH(k, grid, block, lds_bytes_for(nst == 0 || nst == 3 || nst == 4 ? nst : 2), st, p, C, ldc, slab_stride);
    DAE_CHECK_LAUNCH();
    return 0;
}

// can the 160 x 128 producer/consumer kernel run this shape with x~^T as a bit image?  (one round of the chip, the bit words of
// the segment fit the producers' registers, whole 64-deep K tiles)
bool dw_bits_fits(int M, int N, int Bp) {
    if (gemm_init()) return false;
    const int tiles_m = (M + DW_BM - 1) / DW_BM, tiles_n = N / BN, per = (tiles_m + 7) / 8;
    return N % BN == 0 && Bp % 64 == 0 && Bp / 64 <= DWB_MAXKT && 8 * per * tiles_n <= g_cus * g_dw_rounds && g_dw_pc != 0;
}

bool dw_x3_fits(int M, int N, int Bp) {
    if (gemm_init()) return false;
    const int tiles_m = (M + DW_BM - 1) / DW_BM, tiles_n = N / BN, per = (tiles_m + 7) / 8;
    return N % BN == 0 && Bp % 64 == 0 && 8 * per * tiles_n <= g_cus * g_dw_rounds_split && g_dw_pc != 0;
}

// does launch_dw_opt take the 160 x 128 producer/consumer kernel for this shape (the only kernel with a transposed-A form)?
bool dw_pc_taken(int M, int N, int K0, int K1, bool grad_only) {
    if (gemm_init()) return false;
    const int tiles_m = (M + DW_BM - 1) / DW_BM, tiles_n = N / BN, per = (tiles_m + 7) / 8;
    const bool fits = g_dw_pc && N % BN == 0 && K0 % 64 == 0 && K1 % 64 == 0 && 8 * per * tiles_n <= g_cus * g_dw_rounds;
    return fits && (grad_only || g_dw_pc == 2 || g_dw_rounds > 1 || 8 * per * tiles_n > (3 * g_cus) / 4);
}

int launch_dw_opt(int M, int N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0, int K0, const void* A1, int64_t lda1,
                  const void* Bt1, int64_t ldb1, int K1, const OptEpi& e, hipStream_t st, const DwBitsArgs* xa, bool tra) {
    GemmParams p;
    // xa: segment 0's A operand is the bit image (A0 == NULL); fill_params only needs a non-null, aligned placeholder
    {
        const GemmSegDesc segs2[2] = {{xa ? Bt0 : A0, xa ? ldb0 : lda0, Bt0, ldb0, K0}, {A1, lda1, Bt1, ldb1, K1}};
        if (int rc = fill_params_n(p, DAE_BF16, M, N, segs2, 2, 1, BN, tra)) return rc;
    }
    if (int rc = gemm_init()) return rc;
    const bool grad_only = e.opt == DW_GRAD_ONLY;
    if (grad_only) {
        DAE_CHECK_ARG((e.grad || e.grad_lo) && e.ldw >= N && e.ldw % 8 == 0, "dw: gradient-only form needs grad or grad_lo");
    } else {
        DAE_CHECK_ARG(e.W && e.W_lo && e.Wt_lo && e.ldw >= N && e.ldwt >= M && e.ldw % 8 == 0 && e.ldwt % 8 == 0, "dw_opt: bad parameter images");
        DAE_CHECK_ARG(e.opt >= DAE_OPT_SGD && e.opt <= DAE_OPT_ADAM && (e.opt == DAE_OPT_SGD || e.s1) && (e.opt != DAE_OPT_ADAM || e.s2),
                      "dw_opt: optimizer slots missing");
    }
    {   // 160 x 128 tiles, 8-wave producer/consumer: one workgroup per CU in a single round when the tile count fits the chip
        const int tiles_m = (M + DW_BM - 1) / DW_BM, tiles_n = N / BN;
        const int per = (tiles_m + 7) / 8;
        const bool fits = g_dw_pc && K0 % 64 == 0 && K1 % 64 == 0 && 8 * per * tiles_n <= g_cus * g_dw_rounds;
        DAE_CHECK_ARG(!xa || (fits && K0 / 64 <= DWB_MAXKT && xa->xtb && xa->ldxt >= K0 / 32),
                      "dw: the bit-image form of x~^T does not fit this shape (M=%d N=%d Bp=%d)", M, N, K0);
        DAE_CHECK_ARG(!grad_only || fits, "dw: the gradient-only form runs on the 160 x 128 kernel only (M=%d N=%d)", M, N);
        if (fits && (xa || grad_only || g_dw_pc == 2 || g_dw_rounds > 1 || 8 * per * tiles_n > (3 * g_cus) / 4)) {
            typedef void (*dwpc_fn)(GemmParams, OptEpi, int, DwBits);
            static const dwpc_fn pcs[2][5] = {
                {gemm_dw_pc<DAE_OPT_SGD, false>, gemm_dw_pc<DAE_OPT_ADAGRAD, false>, gemm_dw_pc<DAE_OPT_MOMENTUM, false>, gemm_dw_pc<DAE_OPT_ADAM, false>,
                 gemm_dw_pc<DW_GRAD_ONLY, false>},
                {gemm_dw_pc<DAE_OPT_SGD, true>, gemm_dw_pc<DAE_OPT_ADAGRAD, true>, gemm_dw_pc<DAE_OPT_MOMENTUM, true>, gemm_dw_pc<DAE_OPT_ADAM, true>,
                 gemm_dw_pc<DW_GRAD_ONLY, true>}};
            static const dwpc_fn pcs_tr[5] = {gemm_dw_pc<DAE_OPT_SGD, false, false, false, true>, gemm_dw_pc<DAE_OPT_ADAGRAD, false, false, false, true>,
                                              gemm_dw_pc<DAE_OPT_MOMENTUM, false, false, false, true>, gemm_dw_pc<DAE_OPT_ADAM, false, false, false, true>,
                                              gemm_dw_pc<DW_GRAD_ONLY, false, false, false, true>};
            static int pc_rc = [] {
                int rc = 0;
                for (int v = 0; v < 2; ++v)
                    for (dwpc_fn f : pcs[v])
                        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
                for (dwpc_fn f : pcs_tr) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
                return rc;
            }();
            DAE_CHECK_ARG(pc_rc == 0, "dw_pc: hipFuncSetAttribute failed");
            DAE_CHECK_ARG(!tra || !xa, "dw: the transposed-A form takes dense row-major images (no bit image)");
            GemmParams q = p;
            q.tiles_m = tiles_m; q.tiles_n = tiles_n;
            DwBits xb; memset(&xb, 0, sizeof(xb));
            if (xa) {
                xb.xtb = xa->xtb; xb.ldxt = xa->ldxt; xb.nwords = K0 / 32;
                xb.one = host_f2bf(xa->scale);                  // 16-bit image of the scale, round to nearest even (a finite positive factor)
                q.seg[0].A = nullptr;
            }
            DAE_LAUNCH(tra ? pcs_tr[e.opt] : pcs[xa ? 1 : 0][e.opt], dim3(8 * per * tiles_n), dim3(PC_THREADS), DW_LDS, st, q, e, M, xb);
            DAE_CHECK_LAUNCH();
            return 0;
        }
    }
    DAE_CHECK_ARG(!tra, "dw: the transposed-A form exists on the 160 x 128 kernel only (ask dw_pc_taken first)");
    typedef void (*dwo_fn)(GemmParams, OptEpi);
    static const dwo_fn fns[4] = {gemm_dw_opt<DAE_OPT_SGD>, gemm_dw_opt<DAE_OPT_ADAGRAD>, gemm_dw_opt<DAE_OPT_MOMENTUM>, gemm_dw_opt<DAE_OPT_ADAM>};
    constexpr int ldsb = DWO_LDS > lds_bytes_for(2) ? DWO_LDS : lds_bytes_for(2);
    static int attr_rc = [] {
        int rc = 0;
        for (dwo_fn f : fns) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        return rc;
    }();
    DAE_CHECK_ARG(attr_rc == 0, "dw_opt: hipFuncSetAttribute failed");
    static_assert(DAE_OPT_SGD == 0 && DAE_OPT_ADAGRAD == 1 && DAE_OPT_MOMENTUM == 2 && DAE_OPT_ADAM == 3, "optimizer enum order");
    DAE_CHECK_ARG(!xa && !grad_only, "dw: this shape needs the dense x~^T image and the fused-optimizer form");
    DAE_LAUNCH(fns[e.opt], dim3(grid_blocks(p)), dim3(GEMM_THREADS), ldsb, st, p, e);
    DAE_CHECK_LAUNCH();
    return 0;
}

// dW GEMM + optimizer in split-bf16 mode: the contraction runs over up to 5 K segments (x~^T.delta1_hi, x~^T.delta1_lo, delta2^T_hi.h^T_hi,
// delta2^T_hi.h^T_lo, delta2^T_lo.h^T_hi) and the epilogue also writes the lo images of both shadows (e.W_lo2, e.Wt_lo2).  One-round
// 160 x 128 kernel only (the shapes whose tiles fill the chip once); other shapes are refused for now.
int launch_dw_opt_n(int M, int N, const GemmSegDesc* segs_in, int nsegs_in, const OptEpi& e, hipStream_t st, bool pair, bool tra) {
    if (tra) pair = false;        // (the transposed-A form walks plain segments)
    // pair: consecutive non-empty segments that share their A operand (x~^T . [delta1^T_hi ; delta1^T_lo], delta2^T_hi . [h^T_hi ; h^T_lo]) become ONE
    // segment with two B operands (GemmParams::bt2): the kernel streams the A tile once and multiplies it with both B tiles of the stage
    DAE_CHECK_ARG(segs_in && nsegs_in >= 1 && nsegs_in <= GEMM_MAX_SEG, "dw_opt_n: %d K segments (1..%d)", nsegs_in, GEMM_MAX_SEG);
    GemmSegDesc segs[GEMM_MAX_SEG];
    const void* second[GEMM_MAX_SEG];
    int nsegs = 0;
    for (int i = 0; i < nsegs_in; ++i) {
        if (segs_in[i].K == 0) continue;
        if (pair && nsegs > 0 && !second[nsegs - 1] && segs[nsegs - 1].A == segs_in[i].A && segs[nsegs - 1].lda == segs_in[i].lda &&
            segs[nsegs - 1].ldb == segs_in[i].ldb && segs[nsegs - 1].K == segs_in[i].K && ((uintptr_t)segs_in[i].Bt % 16) == 0) {
            second[nsegs - 1] = segs_in[i].Bt;
            continue;
        }
        segs[nsegs] = segs_in[i]; second[nsegs] = nullptr; ++nsegs;
    }
    DAE_CHECK_ARG(nsegs >= 1, "dw_opt_n: every K segment is empty");
    GemmParams p;
    if (int rc = fill_params_n(p, DAE_BF16, M, N, segs, nsegs, 1, BN, tra)) return rc;
    DAE_CHECK_ARG(p.nseg == nsegs, "dw_opt_n: segment bookkeeping");
    bool any_pair = false;
    for (int i = 0; i < nsegs; ++i) { p.bt2[i] = (const char*)second[i]; any_pair = any_pair || second[i]; }
    if (int rc = gemm_init()) return rc;
    const bool grad_only = e.opt == DW_GRAD_ONLY;      // data parallel: the fp32 gradient goes to memory (e.grad), nothing is updated
    if (grad_only) {
        DAE_CHECK_ARG(e.grad && !e.grad_lo && e.ldw >= N && e.ldw % 8 == 0, "dw_opt_n: the gradient-only form of split-bf16 mode writes the fp32 image e.grad");
    } else {
        DAE_CHECK_ARG(e.W && e.W_lo && e.Wt_lo && e.Wt_lo2 && e.ldw >= N && e.ldwt >= M && e.ldw % 8 == 0 && e.ldwt % 8 == 0,
                      "dw_opt_n: bad parameter images (split-bf16 mode needs Wt_lo2; W_lo2 is optional)");
        DAE_CHECK_ARG(e.opt >= DAE_OPT_SGD && e.opt <= DAE_OPT_ADAM && (e.opt == DAE_OPT_SGD || e.s1) && (e.opt != DAE_OPT_ADAM || e.s2),
                      "dw_opt_n: optimizer slots missing");
    }
    const int tiles_m = (M + DW_BM - 1) / DW_BM, tiles_n = N / BN, per = (tiles_m + 7) / 8;
    bool k64 = true;
    for (int i = 0; i < nsegs; ++i) k64 = k64 && segs[i].K % 64 == 0;
    DAE_CHECK_ARG(g_dw_pc && k64 && 8 * per * tiles_n <= g_cus * g_dw_rounds_split,
                  "dw_opt_n: the split-mode dW kernel runs shapes of at most %d 160 x 128 tiles per CU (M=%d N=%d)", g_dw_rounds_split, M, N);
    typedef void (*dwpc_fn)(GemmParams, OptEpi, int, DwBits);
    static const dwpc_fn x3s[2][5] = {{gemm_dw_pc<DAE_OPT_SGD, false, true>, gemm_dw_pc<DAE_OPT_ADAGRAD, false, true>,
                                       gemm_dw_pc<DAE_OPT_MOMENTUM, false, true>, gemm_dw_pc<DAE_OPT_ADAM, false, true>, gemm_dw_pc<DW_GRAD_ONLY, false, true>},
                                      {gemm_dw_pc<DAE_OPT_SGD, false, true, true>, gemm_dw_pc<DAE_OPT_ADAGRAD, false, true, true>,
                                       gemm_dw_pc<DAE_OPT_MOMENTUM, false, true, true>, gemm_dw_pc<DAE_OPT_ADAM, false, true, true>,
                                       gemm_dw_pc<DW_GRAD_ONLY, false, true, true>}};
    static_assert(DW_GRAD_ONLY == 4, "the gradient-only instantiation sits at index 4");
    static const dwpc_fn x3s_tr[5] = {gemm_dw_pc<DAE_OPT_SGD, false, true, false, true>, gemm_dw_pc<DAE_OPT_ADAGRAD, false, true, false, true>,
                                      gemm_dw_pc<DAE_OPT_MOMENTUM, false, true, false, true>, gemm_dw_pc<DAE_OPT_ADAM, false, true, false, true>,
                                      gemm_dw_pc<DW_GRAD_ONLY, false, true, false, true>};
    static int rc3 = [] {
        int rc = 0;
        for (dwpc_fn f : x3s_tr) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        for (dwpc_fn f : x3s[0]) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        for (dwpc_fn f : x3s[1]) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, DW_RING2);
        return rc;
    }();
    DAE_CHECK_ARG(rc3 == 0, "dw_opt_n: hipFuncSetAttribute failed");
    GemmParams q = p;
    q.tiles_m = tiles_m; q.tiles_n = tiles_n;
    DwBits xb; memset(&xb, 0, sizeof(xb));
    DAE_LAUNCH(tra ? x3s_tr[e.opt] : x3s[any_pair ? 1 : 0][e.opt], dim3(8 * per * tiles_n), dim3(PC_THREADS), any_pair ? DW_RING2 : DW_LDS, st, q, e, M, xb);
    DAE_CHECK_LAUNCH();
    return 0;
}

int launch_decode_loss(int dtype, int Bp, int Fp, int Hp, const void* h_lo, int64_t ldh, const void* W_lo, int64_t ldw,
                       const DecodeEpi& e_in, hipStream_t st) {
    const GemmSegDesc seg{h_lo, ldh, W_lo, ldw, Hp};
    return launch_decode_loss_n(dtype, Bp, Fp, &seg, 1, e_in, st);
}

// z2 = sum_s h_s . W_s^T over 1..5 K segments (split-bf16 mode: (h_hi,W_hi) (h_hi,W_lo) (h_lo,W_hi)); with e.delta2_2 / e.delta2_t2 set the
// epilogue also writes the lo images of delta2 / delta2^T
int launch_decode_loss_n(int dtype, int Bp, int Fp, const GemmSegDesc* segs, int nsegs, const DecodeEpi& e_in, hipStream_t st) {
    DecodeEpi e = e_in;
    GemmParams p;
    const bool wide = dtype == DAE_BF16 && e.bn == BN && !(e.delta2_2 || e.delta2_t2 || e.x2);
    DAE_CHECK_ARG(e.bn == 0 || e.bn == decode_tile_n(dtype) || wide, "decode_loss: tile width %d is not available for this mode (lo images of delta2 / x need the 64-column tile)", e.bn);
    const int bn = wide ? BN : decode_tile_n(dtype);
    if (int rc = fill_params_n(p, dtype, Bp, Fp, segs, nsegs, 1, bn)) return rc;
    if (int rc = gemm_init()) return rc;
    DAE_CHECK_ARG(e.dec_act >= 0 && e.dec_act <= 2 && e.loss_func >= 0 && e.loss_func <= 2, "decode_loss: bad act/loss");
    DAE_CHECK_ARG(e.ldx % 8 == 0 && (!e.delta2 || e.ldd % 8 == 0) && (!e.delta2_t || e.lddt % 8 == 0),
                  "decode_loss: leading dimensions must be multiples of 8 elements");
    decode_fn k = dtype == DAE_BF16 ? decode_kernel<bf16_t>(e.loss_func, e.dec_act) : decode_kernel<float>(e.loss_func, e.dec_act);
    if (e.x_bits) {
        DAE_CHECK_ARG(dtype == DAE_BF16 && e.ldxb >= Fp / 32 && ((uintptr_t)e.x_bits % 4) == 0, "decode_loss: bad x bit image");
        k = decode_kernel_xbits(e.loss_func, e.dec_act);
    }
    // two K segments over the same A operand and K extent (the split modes' decode): the paired K loop
    bool paird = false;
    if (g_decode_pair && dtype == DAE_BF16 && !wide && !(e.delta2_2 || e.delta2_t2 || e.x2) && p.nseg == 2 && p.seg[0].A == p.seg[1].A &&
        p.seg[0].lda_b == p.seg[1].lda_b && p.seg[0].ldb_b == p.seg[1].ldb_b && p.seg[0].ktiles == p.seg[1].ktiles) {
        paird = true;
        k = decode_kernel_pair(e.loss_func, e.dec_act, e.x_bits != nullptr);
        static int pair_rc = [] {
            int rc = 0;
            for (int l = 0; l < 3; ++l)
                for (int a = 0; a < 3; ++a)
                    for (int x = 0; x < 2; ++x)
                        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel_pair(l, a, x != 0)),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, DECODE_PAIR_LDS);
            return rc;
        }();
        DAE_CHECK_ARG(pair_rc == 0, "decode_loss: hipFuncSetAttribute failed");
    }
    // three K segments (h_hi, W_hi) (h_hi, W_lo) (h_lo, W_hi) over one K extent: the two-stage walk that keeps the hi fragments in registers
    bool x3d = false;
    if (g_decode_x3 && dtype == DAE_BF16 && e.x_bits && !wide && !paird && !(e.delta2_2 || e.delta2_t2 || e.x2) && p.nseg == 3 && p.seg[0].A == p.seg[1].A &&
        p.seg[0].Bt == p.seg[2].Bt && p.seg[0].A != p.seg[2].A && p.seg[0].Bt != p.seg[1].Bt && p.seg[0].lda_b == p.seg[1].lda_b && p.seg[0].lda_b == p.seg[2].lda_b &&
        p.seg[0].ldb_b == p.seg[1].ldb_b && p.seg[0].ldb_b == p.seg[2].ldb_b && p.seg[0].ktiles == p.seg[1].ktiles && p.seg[0].ktiles == p.seg[2].ktiles) {
        x3d = true;
        k = decode_kernel_x3(e.loss_func, e.dec_act);
        static int x3_rc = [] {
            int rc = 0;
            for (int l = 0; l < 3; ++l)
                for (int a = 0; a < 3; ++a)
                    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel_x3(l, a)), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   DecGeo<DECODE_BN_BF16>::LDS_BYTES);
            return rc;
        }();
        DAE_CHECK_ARG(x3_rc == 0, "decode_loss: hipFuncSetAttribute failed");
    }
    // two K segments over the same h (h.W_hi + h.W_lo), binary input: the register-carry walk (the lo stage is the W_lo tile alone)
    bool c2d = false;
    if (g_decode_x3 && dtype == DAE_BF16 && e.x_bits && !wide && !paird && !x3d && !e.x2 && p.nseg == 2 && p.seg[0].A == p.seg[1].A && p.seg[0].Bt != p.seg[1].Bt &&
        p.seg[0].lda_b == p.seg[1].lda_b && p.seg[0].ldb_b == p.seg[1].ldb_b && p.seg[0].ktiles == p.seg[1].ktiles) {
        c2d = true;
        static int c2_rc = [] {
            int rc = 0;
            for (int l = 0; l < 3; ++l)
                for (int a = 0; a < 3; ++a)
                    for (int rs = 0; rs < 2; ++rs)
                        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel_c2(l, a, rs != 0)), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       DecGeo<DECODE_BN_BF16>::LDS_BYTES);
            return rc;
        }();
        DAE_CHECK_ARG(c2_rc == 0, "decode_loss: hipFuncSetAttribute failed");
    }
    if (e.op_scale == 0.f) e.op_scale = 1.f;
    e.no_pad_skip = g_pad_skip ? 0 : 1;
    e.dbg = g_decode_dbg;
    // A-stationary persistent form: 16-bit, 64-column tiles, no lo images, every K segment over the same h with K <= 8 tiles (Hp <= 512)
    bool ast = g_decode_ast && dtype == DAE_BF16 && !wide && !paird && !x3d && !c2d && !(e.delta2_2 || e.delta2_t2 || e.x2) && p.seg[0].ktiles <= DecAst::MAXKT && p.nseg <= 2;
    for (int s = 1; s < p.nseg && ast; ++s)
        ast = p.seg[s].A == p.seg[0].A && p.seg[s].lda_b == p.seg[0].lda_b && p.seg[s].ldb_b == p.seg[0].ldb_b && p.seg[s].ktiles == p.seg[0].ktiles;
    if (ast) {
        decode_fn ka = decode_kernel_ast(e.loss_func, e.dec_act, e.x_bits != nullptr);
        static int ast_rc = [] {
            int rc = 0;
            for (int l = 0; l < 3; ++l)
                for (int a = 0; a < 3; ++a)
                    for (int x = 0; x < 2; ++x)
                        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel_ast(l, a, x != 0)),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, DecAst::LDS_BYTES);
            return rc;
        }();
        DAE_CHECK_ARG(ast_rc == 0, "decode_loss: hipFuncSetAttribute failed");
        if (e.sym_G) DAE_CHECK_ARG(e.sym_scalars && e.sym_Gs && e.sym_Bp % 64 == 0 && e.sym_B <= e.sym_Bp, "decode_loss: bad sym_scale rider");
        const int tiles = p.tiles_m * p.tiles_n;
        int nwg = 2 * (g_cus > 0 ? g_cus : 256);          // two resident workgroups per CU
        if (nwg > tiles) nwg = tiles;
        DAE_LAUNCH(ka, dim3(nwg), dim3(GEMM_THREADS), DecAst::LDS_BYTES, st, p, e);
        DAE_CHECK_LAUNCH();
        return 0;
    }
    if (wide) {
        k = decode_kernel_wide(e.loss_func, e.dec_act, e.x_bits != nullptr);
        static int wide_rc = [] {
            int rc = 0;
            for (int l = 0; l < 3; ++l)
                for (int a = 0; a < 3; ++a)
                    for (int x = 0; x < 2; ++x)
                        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel_wide(l, a, x != 0)),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, DecGeo<BN>::LDS_BYTES);
            return rc;
        }();
        DAE_CHECK_ARG(wide_rc == 0, "decode_loss: hipFuncSetAttribute failed");
    }
    if (e.delta2_2 || e.delta2_t2 || e.x2) {
        DAE_CHECK_ARG(dtype == DAE_BF16, "decode_loss: lo images of delta2 / x exist in the 16-bit split mode only");
        k = decode_kernel_res(e.loss_func, e.dec_act, e.x_bits != nullptr);
        static int res_rc = [] {
            int rc = 0;
            for (int l = 0; l < 3; ++l)
                for (int a = 0; a < 3; ++a)
                    for (int x = 0; x < 2; ++x)
                        rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel_res(l, a, x != 0)),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, DecGeo<DECODE_BN_BF16>::LDS_BYTES);
            return rc;
        }();
        DAE_CHECK_ARG(res_rc == 0, "decode_loss: hipFuncSetAttribute failed");
    }
    if (c2d) k = decode_kernel_c2(e.loss_func, e.dec_act, e.delta2_2 || e.delta2_t2);
    int nblocks = grid_blocks(p);
    if (e.sym_G) {
        DAE_CHECK_ARG(e.sym_scalars && e.sym_Gs && e.sym_Bp % 64 == 0 && e.sym_B <= e.sym_Bp, "decode_loss: bad sym_scale rider");
        e.sym_first = nblocks;
        nblocks += (e.sym_Bp / 64) * (e.sym_Bp / 64);
    }
    dim3 grid(nblocks), block(GEMM_THREADS);
    static_assert(DecGeo<DECODE_BN_BF16>::LDS_BYTES >= 64 * 65 * 4 && DecGeo<BN>::LDS_BYTES >= 64 * 65 * 4, "rider tile must fit the decode LDS");
    DAE_LAUNCH(k, grid, block, paird ? DECODE_PAIR_LDS : ((dtype == DAE_BF16 && !wide) ? DecGeo<DECODE_BN_BF16>::LDS_BYTES : DecGeo<BN>::LDS_BYTES), st, p, e);
    DAE_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Fused corrupt+encode GEMM for BINARY inputs (autoencoder.py:389: tf.sparse.matmul(x~, W) on a 0/1 CSR):
//   z1 slab[split] = bits(x~)[Bp x Fp] . Wt_lo[Hp x Fp]^T
// The corrupted batch reaches the kernel as the gather's BIT image (1 bit per feature: 1.1 MB instead of 18 MB of bf16);
// the dense x~ operand is never written to or read from HBM.  Producer/consumer structure of gemm_nt_pc:
//   consumer waves 0-3: fragment reads + MFMAs, byte-for-byte the loop of gemm_nt_pc (the LDS image is identical);
//   producer waves 4-7: W^T tile by LDS-DMA (4 x 1 KiB pieces per wave per K tile -- HALF the DMA instructions of the dense
//     kernel, whose ~70-cycle issue cost per piece is what paces its K loop), and the x~ tile BUILT in LDS from the bits:
//     zero-fill of the wave's 32 rows (4 ds_write_b128) + one ds_write_b16 of bf16 1.0 per set bit (x~ is ~1.4 % dense:
//     ~0.5 set bits per lane and K tile); words with many bits (dense rows, salt-and-pepper) are expanded arithmetically.
// The workgroup's slice of the bit image (128 rows x 2 words per K tile) is copied into LDS once, before the K loop, so the
// loop itself issues no ordinary global load (hipcc drains the LDS-DMA queue at every use of one).
// ------------------------------------------------------------------------------------------------
constexpr int eb_slice_stride(int nk) { return 2 * ((nk | 1)); }            // words per row: 2 * odd >= 2 nk -> conflict-free ds_read_b32
constexpr int eb_lds_bytes(int nst, int nk) { return nst * STAGE_BYTES + 128 * eb_slice_stride(nk) * 4; }
static int eb_ring_depth(int nk) { return eb_lds_bytes(4, nk) <= 160 * 1024 ? 4 : (eb_lds_bytes(2, nk) <= 160 * 1024 ? 2 : 0); }

struct EncBitsParams {
    const char* Bt; int64_t ldb_b;                    // W^T_lo [Hp x Fp] bf16, leading dimension in bytes
    const uint32_t* bits; int64_t ldw;                // x~ bits [Bp x ldw words], bit b of word w = feature 32*w + b
    int ktiles_total, tiles_m, tiles_n, splits, nk_max;
};

template <int NST>
__global__ __launch_bounds__(PC_THREADS, 1) void gemm_encode_bits_pc(EncBitsParams p, float* __restrict__ C, int64_t ldc, int64_t slab_stride,
                                                                     LabelJob job, int label_block) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if ((int)blockIdx.x == label_block) { label_stats_block<PC_THREADS>(job, lds); return; }
    const int id = blockIdx.x;
    const int split = id % p.splits, tile = id / p.splits;          // a K slice stays on one XCD (block b runs on XCD b % 8)
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int kt0 = (int)(((int64_t)p.ktiles_total * split) / p.splits);
    const int kt1 = (int)(((int64_t)p.ktiles_total * (split + 1)) / p.splits);
    const int nk = kt1 - kt0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0_m = tm * BM, row0_n = tn * BN;
    uint32_t* const slice = reinterpret_cast<uint32_t*>(lds + NST * STAGE_BYTES);
    const int sstride = eb_slice_stride(p.nk_max);
    // ---- the workgroup's slice of the bit image -> LDS (all 8 waves; 4 threads per row) ----
    {
        const int row = tid >> 2, part = tid & 3;
        const uint32_t* src = p.bits + (int64_t)(row0_m + row) * p.ldw + 2 * kt0;
        for (int w = part; w < 2 * nk; w += 4) slice[row * sstride + w] = src[w];
    }

    if (wave8 >= 4) {
        // ================= producer =================
        if (nk <= 0) { __builtin_amdgcn_s_barrier(); return; }
        const int wave = wave8 - 4;
        uint32_t voB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 4 + wave) * 8 + (lane >> 3);
            voB[i] = (uint32_t)(row0_n + row) * (uint32_t)p.ldb_b + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        const char* gB = p.Bt + (int64_t)kt0 * BKB;
        auto dma_stage = [&](char* slot) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + voB[i]),
                                                 (__attribute__((address_space(3))) void*)(slot + TILE_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
            gB += BKB;
        };
        // A tile of K tile t (relative to kt0) into `slot`: this wave's rows [32 wave, +32); lane = (row, 32-feature half)
        const int arow = wave * 32 + (lane >> 1), half = lane & 1;
        const uint32_t aswz = (uint32_t)((arow >> 1) & 7);
        const uint32_t* const myword = slice + arow * sstride + half;
        auto build_a = [&](int t, char* slot) {
            uint32_t word = myword[2 * t];
            const bool dense = __builtin_amdgcn_ballot_w64(__builtin_popcount(word) > 6) != 0ull;
            if (dense) {                                   // arithmetic expansion: 8 bits -> 8 bf16 (0 / 1.0) per 16-byte slot
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    i32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t b2 = (word >> (8 * j + 2 * q)) & 3u;
                        v[q] = (int)(((b2 & 1u) * kOne16) | ((b2 >> 1) * (kOne16 << 16)));
                    }
                    *reinterpret_cast<i32x4*>(slot + arow * BKB + (((uint32_t)(half * 4 + j) ^ aswz) << 4)) = v;
                }
            } else {
                const i32x4 z = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<i32x4*>(slot + (wave * 4 + i) * 1024 + lane * 16) = z;
                while (word) {                             // one 2-byte store per set bit (in-order LDS: lands after the zero fill)
                    const int b = __builtin_ctz(word);
                    word &= word - 1;
                    const uint32_t k = (uint32_t)(half * 32 + b);
                    *reinterpret_cast<bf16_t*>(slot + arow * BKB + (((k >> 3) ^ aswz) << 4) + (k & 7) * 2) = (bf16_t)kOne16;
                }
            }
        };
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // my part of the bit slice is in LDS
        __builtin_amdgcn_s_barrier();                                     // ... and everybody else's
        asm volatile("" ::: "memory");
#pragma unroll
        for (int st = 0; st < NST; ++st)
            if (st < nk) { dma_stage(lds + st * STAGE_BYTES); build_a(st, lds + st * STAGE_BYTES); }
        if (nk >= NST) wait_vm<(NST - 1) * 4>(); else wait_vm<0>();      // stage 0's W^T tile landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // ... and every A tile written so far
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int i = 0; i < nk; ++i) {
            const int ahead = min(NST - 2, nk - 2 - i);                   // stages younger than i+1 already requested (4 DMAs each)
            if (ahead >= 2) wait_vm<8>();
            else if (ahead == 1) wait_vm<4>();
            else wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (i + NST < nk) { dma_stage(lds + cur * STAGE_BYTES); build_a(i + NST, lds + cur * STAGE_BYTES); }
            cur = cur + 1 == NST ? 0 : cur + 1;
        }
        return;
    }

    // ================= consumer (same loop as gemm_nt_pc) =================
    const int wave = wave8;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // my part of the bit slice is in LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nk > 0) {
        const int r = lane & 31, g = lane >> 5;
        const int swz = (r >> 1) & 7;
        const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
        const uint32_t offa = (wm * 64 + r) * BKB, offb = TILE_BYTES + (wn * 64 + r) * BKB;
        uint32_t so[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
        i32x4 fa[4][2], fb[4][2];
#define DAE_READ_KK(KK, SLOTBASE)                                          \
    fa[KK][0] = lds_read_b128((SLOTBASE) + offa + so[KK]);                 \
    fa[KK][1] = lds_read_b128_off4096((SLOTBASE) + offa + so[KK]);         \
    fb[KK][0] = lds_read_b128((SLOTBASE) + offb + so[KK]);                 \
    fb[KK][1] = lds_read_b128_off4096((SLOTBASE) + offb + so[KK]);
#define DAE_MMA4(KK)                                                       \
    Mma<bf16_t>::run(fa[KK][0], fb[KK][0], acc[0][0]);                     \
    Mma<bf16_t>::run(fa[KK][0], fb[KK][1], acc[0][1]);                     \
    Mma<bf16_t>::run(fa[KK][1], fb[KK][0], acc[1][0]);                     \
    Mma<bf16_t>::run(fa[KK][1], fb[KK][1], acc[1][1]);
        __builtin_amdgcn_s_barrier();                                     // stage 0 complete (producers waited for it)
        asm volatile("" ::: "memory");
        DAE_READ_KK(0, lbase) DAE_READ_KK(1, lbase) DAE_READ_KK(2, lbase) DAE_READ_KK(3, lbase)
        __builtin_amdgcn_sched_barrier(0);
        int cur = 0;
        for (int i = 0; i < nk; ++i) {
            const int nxt = cur + 1 == NST ? 0 : cur + 1;
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");            // R0 (kk 0,1) of tile i
            __builtin_amdgcn_sched_barrier(0);
            DAE_MMA4(0) DAE_MMA4(1)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // R1 landed; every LDS read of tile i is done
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint32_t nb = lbase + nxt * STAGE_BYTES;
            DAE_READ_KK(0, nb) DAE_READ_KK(1, nb)                         // stale (never consumed) after the last tile
            __builtin_amdgcn_sched_barrier(0);
            DAE_MMA4(2) DAE_MMA4(3)
            __builtin_amdgcn_sched_barrier(0);
            DAE_READ_KK(2, nb) DAE_READ_KK(3, nb)
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef DAE_READ_KK
#undef DAE_MMA4
    }
    const int g = lane >> 5, c = lane & 31;
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

int launch_encode_bits(int Bp, int Hp, int Fp, const uint32_t* bits, int64_t ldw, const void* Wt_lo, int64_t ldb, float* C,
                       int64_t ldc, int splits, int64_t slab_stride, hipStream_t st, const LabelJob* label_job, int* label_done) {
    if (label_done) *label_done = 0;
    DAE_CHECK_ARG(bits && Wt_lo && C, "encode_bits: null operand");
    DAE_CHECK_ARG(Bp % BM == 0 && Hp % BN == 0 && Fp % 64 == 0 && ldw >= Fp / 32, "encode_bits: bad shape");
    DAE_CHECK_ARG(((uintptr_t)Wt_lo % 16) == 0 && (ldb * 2) % 16 == 0 && ((uintptr_t)bits % 4) == 0, "encode_bits: alignment");
    DAE_CHECK_ARG((uint64_t)Hp * (uint64_t)ldb * 2 < (1ull << 32), "encode_bits: W^T panel must stay below 4 GiB (32-bit DMA offsets)");
    if (int rc = gemm_init()) return rc;
    EncBitsParams p;
    p.Bt = (const char*)Wt_lo; p.ldb_b = ldb * 2; p.bits = bits; p.ldw = ldw;
    p.ktiles_total = Fp / 64; p.tiles_m = Bp / BM; p.tiles_n = Hp / BN; p.splits = splits < 1 ? 1 : splits;
    DAE_CHECK_ARG(p.splits <= p.ktiles_total, "encode_bits: too many splits");
    p.nk_max = (p.ktiles_total + p.splits - 1) / p.splits;
    const int nst = eb_ring_depth(p.nk_max);
    const int grid = p.tiles_m * p.tiles_n * p.splits;
    DAE_CHECK_ARG(nst != 0, "encode_bits: %d K tiles per slice do not fit the LDS: raise the number of splits", p.nk_max);
    const int ldsb = eb_lds_bytes(nst, p.nk_max);
    static int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_encode_bits_pc<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) |
                         (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_encode_bits_pc<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    DAE_CHECK_ARG(attr_rc == 0, "encode_bits: hipFuncSetAttribute failed (%d)", attr_rc);
    LabelJob job; memset(&job, 0, sizeof(job));
    const bool with_labels = label_job && label_job->Bp <= 1024 && grid < g_cus;
    if (with_labels) job = *label_job;
    auto k = nst == 4 ? gemm_encode_bits_pc<4> : gemm_encode_bits_pc<2>;
    DAE_LAUNCH(k, dim3(grid + (with_labels ? 1 : 0)), dim3(PC_THREADS), ldsb, st, p, C, ldc, slab_stride, job, with_labels ? grid : -1);
    DAE_CHECK_LAUNCH();
    if (with_labels && label_done) *label_done = 1;
    return 0;
}

// can the 8-wave bit-image encode kernel run this shape?  (one workgroup per CU; the K slice's bits + the ring fit the LDS)
bool encode_bits_fits(int Bp, int Hp, int Fp, int splits) {
    if (gemm_init()) return false;
    const int kt = Fp / 64;
    if (splits < 1 || splits > kt || Fp % 64 || Bp % BM || Hp % BN) return false;
    return (Bp / BM) * (Hp / BN) * splits <= g_cus && eb_ring_depth((kt + splits - 1) / splits) == 4;
}

int launch_gemm_trace(int dtype, int M, int N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0, int K0, const void* A1,
                      int64_t lda1, const void* Bt1, int64_t ldb1, int K1, float* C, int64_t ldc, int splits, int64_t slab_stride,
                      int nst, unsigned long long* trace, hipStream_t st) {
    GemmParams p;
    if (int rc = fill_params(p, dtype, M, N, A0, lda0, Bt0, ldb0, K0, A1, lda1, Bt1, ldb1, K1, splits)) return rc;
    DAE_CHECK_ARG(trace && dtype == DAE_BF16 && (nst == 2 || nst == 3), "gemm_trace: bf16, nst 2 or 3, trace buffer required");
    p.trace = trace;
    f32out_fn k = nst == 3 ? gemm_nt_trace<bf16_t, 3> : gemm_nt_trace<bf16_t, 2>;
    DAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_for(nst)));
    DAE_LAUNCH(k, dim3(grid_blocks(p)), dim3(GEMM_THREADS), lds_bytes_for(nst), st, p, C, ldc, slab_stride);
    DAE_CHECK_LAUNCH();
    return 0;
}

void set_use_glds(int nst) {
    if (nst == -1) { g_use_pc = 0; return; }          // A/B: 4-wave kernel for every grid
    if (nst == -2) { g_use_pc = 1; return; }
    if (nst == -3) { g_dw_pc = 0; return; }          // A/B: dW on the 4-wave 128 x 128 kernel
    if (nst == -4) { g_dw_pc = 1; return; }
    if (nst == -5) { g_dw_pc = 2; return; }          // tests: the 160 x 128 kernel for every grid that fits one round
    if (nst == -6) { g_w8 = 0; return; }             // A/B: never the 256 x 256 / 8-MFMA-wave kernel
    if (nst == -7) { g_w8 = 1; return; }
    if (nst <= -500000 && nst > -500128) { g_decode_dbg = -500000 - nst; return; }
    if (nst <= -100 && nst > -1000) { g_dw_rounds = g_dw_rounds_split = (-nst - 100 < 1 ? 1 : -nst - 100); return; }    // rounds of the chip the 160 x 128 dW kernel may take
    if (nst == -13) { g_decode_pair = 0; return; }
    if (nst == -14) { g_decode_pair = 1; return; }
    if (nst == -19) { g_gram_fused = 0; return; }
    if (nst == -20) { g_gram_fused = 1; return; }
    if (nst == -17) { g_decode_x3 = 0; return; }
    if (nst == -18) { g_decode_x3 = 1; return; }
    if (nst == -15) { g_decode_ast = 0; return; }
    if (nst == -16) { g_decode_ast = 1; return; }
    if (nst == -11) { g_pad_skip = 0; return; }
    if (nst == -12) { g_pad_skip = 1; return; }
    if (nst == -9) { g_pc_vec = 0; return; }         // A/B: gemm_nt_pc stores its tile as dwords straight from the accumulators
    if (nst == -10) { g_pc_vec = 1; return; }
    if (nst <= -1000) {                              // tests: pretend the device has (-nst - 1000) compute units, so that every CU-count-keyed
        if (gemm_init() == 0) g_cus = -nst - 1000;   // dispatch (one-round kernels, label riders, 256 x 256 slices) is exercised on any box
        return;
    }
    g_nst = nst;
}

}  // namespace dae
