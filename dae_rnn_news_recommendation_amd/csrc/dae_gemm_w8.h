// dae_gemm_w8.h -- 256 x 256 tile NT GEMM for the large contractions (dense-input configs: x~[B x F].W, delta2.W with F = 50 000),
// bf16 operands, fp32 split-K slabs out.  Included by dae_gemm.hip (shares its GemmParams / LDS image) and by tools/gemm_probe.hip.
//
// Why another main loop: the 128 x 128 kernels (gemm_mainloop / gemm_nt_pc) keep ONE MFMA wave per SIMD -- every lgkmcnt / barrier
// stall of that wave is exposed, and a 64 x 64 wave tile reads 16 fragments for 16 MFMAs (LDS read bound at 1:1).  Here
//   * 8 waves ALL issue MFMAs (2 per SIMD: while one waits at the barrier or for its fragments, the other feeds the matrix pipe),
//   * wave tile 128 x 64 (4 x 2 accumulators of 32 x 32): 6 fragment reads per 8 MFMAs, 94 B/clk of LDS reads per CU,
//   * one workgroup per CU, 2 LDS stages of 64 KiB (A 256 x 128 B + B 256 x 128 B), filled by global_load_lds_dwordx4 a whole
//     K tile ahead (the DMA of tile i+2 is issued right after the barrier that retires tile i, so vmcnt(0) before the next
//     barrier never waits in steady state),
//   * fragments double buffered per 16-deep k chunk: chunk kk+1 is read while the 8 MFMAs of chunk kk run (counted lgkmcnt),
//   * ONE s_barrier per K tile (64 deep): [all my reads of tile i done] + [my DMA pieces of tile i+1 landed] -> barrier ->
//     DMA(tile i+2 -> slot of tile i), first chunk of tile i+1, MFMAs of the last chunk of tile i.
// LDS image of a stage: the one of dae_gemm.hip -- rows of 128 B, eight 16-byte slots XOR-swizzled with (row >> 1) & 7, written
// lane-linear by the DMA (the swizzle sits on the per-lane SOURCE address), conflict-free ds_read_b128.
// Rows of the last row tile beyond M re-read row M-1 and are never stored (B = 800 -> Bp = 896 = 3.5 tiles).
#pragma once

namespace dae {

constexpr int W8_BM = 256, W8_BN = 256;
constexpr int W8_THREADS = 512;
constexpr int W8_TILE_BYTES = W8_BM * BKB;          // 32 KiB per operand per stage
constexpr int W8_STAGE = 2 * W8_TILE_BYTES;         // 64 KiB
constexpr int W8_LDS = 2 * W8_STAGE;                // 128 KiB

struct W8Params {
    GemmSeg seg[GEMM_MAX_SEG];
    int nseg;
    int ktiles_total;
    int M, N;                 // valid rows of A / of Bt (multiples of 32; tiles are clamped to them)
    int tiles_m, tiles_n, splits;
};

// block -> (tile, K slice): slice = b % splits (splits % 8 == 0 -> one XCD per slice, its operand slabs stay in that XCD's L2)
__device__ __forceinline__ void w8_block_to_tile(const W8Params& p, int& tm, int& tn, int& split, int& kt0, int& kt1) {
    const int id = blockIdx.x;
    split = id % p.splits;
    const int tile = id / p.splits;
    tn = tile % p.tiles_n;
    tm = tile / p.tiles_n;
    kt0 = (int)(((int64_t)p.ktiles_total * split) / p.splits);
    kt1 = (int)(((int64_t)p.ktiles_total * (split + 1)) / p.splits);
}

// K loop of one 256 x 256 tile: rows [row0_m, +256) of A (clamped to p.M), rows [row0_n, +256) of Bt (clamped to p.N), K tiles [kt0, kt1).
// acc[mt][nt]: the wave's 128 x 64 block (wm = wave >> 2, wn = wave & 3), 32 x 32 accumulators in the MFMA register layout.
__device__ __forceinline__ void w8_mainloop(const W8Params& p, int row0_m, int row0_n, int kt0, int kt1, char* lds, f32x16 (&acc)[4][2]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // 2 x 4 waves: rows [128 wm, +128), columns [64 wn, +64)
    const int nk = kt1 - kt0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (nk <= 0) return;
    // ---- LDS-DMA addressing: 4 pieces (8 rows x 128 B) per operand per wave and stage ----
    uint32_t voA[4], voB[4];
    const char *gA = nullptr, *gB = nullptr;
    int kt_dma = kt0, seg_end = 0;
    auto seg_setup = [&](int kt) {
        int k;
        const int sg = seg_locate(p, kt, k, seg_end);
        const uint32_t lda = (uint32_t)p.seg[sg].lda_b, ldb = (uint32_t)p.seg[sg].ldb_b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 8 + wave) * 8 + (lane >> 3);
            const uint32_t ss = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
            voA[i] = (uint32_t)min(row0_m + row, p.M - 1) * lda + ss;
            voB[i] = (uint32_t)min(row0_n + row, p.N - 1) * ldb + ss;
        }
        gA = p.seg[sg].A + (int64_t)k * BKB;
        gB = p.seg[sg].Bt + (int64_t)k * BKB;
    };
    seg_setup(kt0);
    auto dma_stage = [&](char* slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = i * 8 + wave;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + voA[i]),
                                             (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + voB[i]),
                                             (__attribute__((address_space(3))) void*)(slot + W8_TILE_BYTES + piece * 1024), 16, 0, 0);
        }
        ++kt_dma;
        if (kt_dma == seg_end) { if (kt_dma < p.ktiles_total) seg_setup(kt_dma); }
        else { gA += BKB; gB += BKB; }
    };

    // ---- fragment addressing ----
    const int r = lane & 31, g = lane >> 5;
    const int swz = (r >> 1) & 7;
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const uint32_t offa = (wm * 128 + r) * BKB, offb = W8_TILE_BYTES + (wn * 64 + r) * BKB;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = (uint32_t)(((kk * 2 + g) ^ swz) << 4);
    i32x4 fa[2][4], fb[2][2];
    // one 16-deep chunk: A fragments of the 4 row blocks (32 rows = 4096 B apart), B fragments of the 2 column blocks
#define W8_READ(SET, KK, SLOTBASE)                                                         \
    asm volatile("ds_read_b128 %0, %1" : "=&v"(fa[SET][0]) : "v"((SLOTBASE) + offa + so[KK]));               \
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(fa[SET][1]) : "v"((SLOTBASE) + offa + so[KK]));   \
    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=&v"(fa[SET][2]) : "v"((SLOTBASE) + offa + so[KK]));   \
    asm volatile("ds_read_b128 %0, %1 offset:12288" : "=&v"(fa[SET][3]) : "v"((SLOTBASE) + offa + so[KK]));  \
    asm volatile("ds_read_b128 %0, %1" : "=&v"(fb[SET][0]) : "v"((SLOTBASE) + offb + so[KK]));               \
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(fb[SET][1]) : "v"((SLOTBASE) + offb + so[KK]));
#define W8_MMA(SET)                                                                        \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                     \
        Mma<bf16_t>::run(fa[SET][mt], fb[SET][0], acc[mt][0]);                             \
        Mma<bf16_t>::run(fa[SET][mt], fb[SET][1], acc[mt][1]);                             \
    }                                                                                      \
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: tiles 0 and 1 requested, tile 0 landed, its first chunk read ----
    dma_stage(lds);
    if (nk > 1) dma_stage(lds + W8_STAGE);
    if (nk > 1) wait_vm<8>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    W8_READ(0, 0, lbase)
    __builtin_amdgcn_sched_barrier(0);
    int cur = 0;
    for (int i = 0; i < nk; ++i) {
        const uint32_t cb = lbase + cur * W8_STAGE, nb = lbase + (cur ^ 1) * W8_STAGE;
        // chunk 0 (set 0) | prefetch chunk 1 -> set 1
        W8_READ(1, 1, cb)
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        W8_MMA(0)
        // chunk 1 (set 1) | prefetch chunk 2 -> set 0
        W8_READ(0, 2, cb)
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        W8_MMA(1)
        // chunk 2 (set 0) | prefetch chunk 3 -> set 1
        W8_READ(1, 3, cb)
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        W8_MMA(0)
        // every read of tile i has been issued; retire them and my DMA pieces of tile i+1, then the tile barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + 2 < nk) dma_stage(lds + cur * W8_STAGE);      // tile i+2 into the slot tile i just left
        W8_READ(0, 0, nb)                                      // first chunk of tile i+1 (stale and unused after the last tile)
        __builtin_amdgcn_sched_barrier(0);
        W8_MMA(1)                                              // chunk 3
        cur ^= 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef W8_READ
#undef W8_MMA
}

template <int ROLE>
__global__ __launch_bounds__(W8_THREADS, 1) void gemm_nt_w8(W8Params p, float* __restrict__ C, int64_t ldc, int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int tm, tn, split, kt0, kt1;
    w8_block_to_tile(p, tm, tn, split, kt0, kt1);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int row0_m = tm * W8_BM, row0_n = tn * W8_BN;
    f32x16 acc[4][2];
    w8_mainloop(p, row0_m, row0_n, kt0, kt1, lds, acc);
    // ---- epilogue: the wave's 128 x 64 block of the slab ----
    const int g = lane >> 5, c = lane & 31;
    float* Cs = C + (int64_t)split * slab_stride;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0_m + wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                const int col = row0_n + wn * 64 + nt * 32 + c;
                if (row < p.M && col < p.N) Cs[(int64_t)row * ldc + col] = acc[mt][nt][r];
            }
}

}  // namespace dae
