// lds_stream_ubench.hip -- how many bytes per second can ONE CU pull out of L2 with vector-memory instructions (gfx950)?
// Stand-alone tool (not part of libdae_hip.so):  hipcc --offload-arch=gfx950 -O3 tools/lds_stream_ubench.hip -o tools/lds_stream_ubench
// The GEMM K loops of this library move 30-42 GB/s per CU through `global_load_lds_dwordx4`, far below what the MFMA pipe could
// consume; this probe separates the candidates: instruction form (LDS-DMA / register dwordx4 / register dword), waves issuing per CU,
// pieces in flight per wave, and whether the CUs read private windows (A panels) or one shared window (the B panel).
// One workgroup per CU; every wave streams `iters` x 16 wave-instructions over a window that stays L2-resident.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
enum Mode { M_DMA16 = 0, M_REG16, M_REG4, M_DMA4, M_COUNT };
static const char* kModes[M_COUNT] = {"global_load_lds_dwordx4 (1 KiB / instr)", "global_load_dwordx4 -> VGPR (1 KiB / instr)",
                                      "global_load_dword -> VGPR (256 B / instr)", "global_load_lds_dword (256 B / instr)"};

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(1024) void stream(const char* __restrict__ src, long long window, long long cu_stride, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int W = (MODE == M_DMA16 || MODE == M_REG16) ? 16 : 4;          // bytes per lane
    constexpr int PIECE = 64 * W;
    const char* base = src + (long long)blockIdx.x * cu_stride;
    char* ring = lds + wave * (16 * PIECE);                                  // 16 pieces per wave
    long long off0 = ((long long)wave * 16 * PIECE) % window;                // waves interleave 16-piece groups over the window
    if (off0 + 16 * PIECE > window) off0 = 0;
    long long off = off0;
    const long long step = (long long)nw * 16 * PIECE;
    i32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const char* g = base + off + lane * W;
        if constexpr (MODE == M_DMA16 || MODE == M_DMA4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (MODE == M_DMA16)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * PIECE),
                                                     (__attribute__((address_space(3))) void*)(ring + i * PIECE), 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * PIECE),
                                                     (__attribute__((address_space(3))) void*)(ring + i * PIECE), 4, 0, 0);
                if (i >= INFLIGHT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
            }
        } else if constexpr (MODE == M_REG16) {          // 16 loads into 16 distinct register quads, drained once per iteration
            i32x4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[i]) : "v"(g + i * PIECE) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += v[i];
        } else {
            int v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("global_load_dword %0, %1, off" : "=&v"(v[i]) : "v"(g + i * PIECE) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) acc.x += v[i];
        }
        off += step;
        if (off + 16 * PIECE > window) off = off0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x + acc.y + acc.z + acc.w == 0x7fffffff) sink[0] = lds[threadIdx.x];
}

// GEMM-tile fetch pattern: one wave-instruction lands 8 rows x 128 B (the K-tile columns [kb*128, +128) of 8 consecutive operand rows
// at row stride S), as the kernels of this library issue it.  R = 64 rows per CU window (8 groups of 8 rows), K tiles swept in order.
template <int WAVES>
__global__ __launch_bounds__(1024) void stream_rows(const char* __restrict__ src, long long S, long long cu_stride, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int GPW = 8 / WAVES;                        // row groups per wave (2 at 4 waves, 1 at 8)
    constexpr int KPI = 16 / GPW;                         // K tiles per iteration of 16 pieces
    const char* base = src + (long long)blockIdx.x * cu_stride;
    char* ring = lds + wave * (16 * 1024);
    const int nkb = (int)(S / 128);
    int kb = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int k = kb + i / GPW; if (k >= nkb) k -= nkb;
            const int grp = wave * GPW + (i % GPW);
            const char* g = base + (long long)(grp * 8 + (lane >> 3)) * S + (long long)k * 128 + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ring + i * 1024), 16, 0, 0);
            if (i >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
        kb += KPI; if (kb >= nkb) kb -= nkb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (iters < 0) sink[0] = lds[threadIdx.x];
}

template <int WAVES>
static void run_rows(const char* src, long long S, bool shared, int cus, int* sink) {
    const int iters = 2048 / WAVES * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream_rows<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("  8 rows x 128 B per instr, row stride %5lld B, 64 rows  %2d waves/CU  15 in flight/wave  %s window %4lld KiB: ", S, WAVES,
           shared ? "shared " : "private", (64 * S) >> 10);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stream_rows<WAVES>), dim3(cus), dim3(64 * WAVES), (size_t)WAVES * 16 * 1024, 0, src, S, shared ? 0LL : 64 * S, iters, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes_cu = (double)iters * 16 * 1024 * WAVES, instr_cu = (double)iters * 16 * WAVES;
    printf("%7.1f GB/s per CU  %6.2f TB/s chip  %6.1f ns per wave-instr per CU  (%s)\n", bytes_cu / best * 1e-6, bytes_cu * cus / best * 1e-9,
           best * 1e6 / instr_cu, hipGetErrorString(hipGetLastError()));
}

template <int MODE, int INFLIGHT>
static void run(const char* src, long long window, bool shared, int waves, int cus, int* sink, bool once = false) {
    constexpr int W = (MODE == M_DMA16 || MODE == M_REG16) ? 16 : 4;
    const int iters = once ? (int)(window / ((long long)waves * 16 * 64 * W)) : 2048 / waves * 4;
    const size_t ldsb = (MODE == M_DMA16 || MODE == M_DMA4) ? (size_t)waves * 16 * 64 * W : 0;
    if (ldsb > 160 * 1024) return;
    printf("  %-44s %2d waves/CU  %2d in flight/wave  %s window %4lld KiB: ", kModes[MODE], waves, INFLIGHT, shared ? "shared " : "private", window >> 10);
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream<MODE, INFLIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stream<MODE, INFLIGHT>), dim3(cus), dim3(64 * waves), ldsb, 0, src, window, shared ? 0LL : window, iters, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes_cu = (double)iters * 16 * 64 * W * waves, instr_cu = (double)iters * 16 * waves;
    printf("%7.1f GB/s per CU  %6.2f TB/s chip  %6.1f ns per wave-instr per CU  (%s)\n", bytes_cu / best * 1e-6, bytes_cu * cus / best * 1e-9,
           best * 1e6 / instr_cu, hipGetErrorString(hipGetLastError()));
}

// Stores, every byte written once (HBM-bound at best).  Form 0: 16 B per lane, the wave-instruction covers 1 KiB of one row; form 1: 4 B
// per lane, 256 contiguous bytes; form 2: 4 B per lane in the MFMA accumulator layout -- lanes 0-31 write 128 B of row r, lanes 32-63
// 128 B of row r + 4 (row stride S bytes), the next instruction moves one row down (what a GEMM epilogue does without staging).
template <int FORM>
__global__ __launch_bounds__(1024) void stream_store(char* __restrict__ dst, long long per_cu, long long S) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    char* base = dst + (long long)blockIdx.x * per_cu;
    if constexpr (FORM == 0) {
        const i32x4 v = {lane, wave, 1, 2};
        for (long long off = (long long)wave * 1024; off + 1024 <= per_cu; off += (long long)nw * 1024)
            *reinterpret_cast<i32x4*>(base + off + lane * 16) = v;
    } else if constexpr (FORM == 1) {
        for (long long off = (long long)wave * 256; off + 256 <= per_cu; off += (long long)nw * 256)
            *reinterpret_cast<int*>(base + off + lane * 4) = lane;
    } else {
        // tiles of 32 rows x 128 B: a wave owns tile t = wave, wave + nw, ...; tile t sits at rows [32 (t / nt_row), +32), byte column 128 (t % nt_row)
        const int nt_row = (int)(S / 128);
        const long long ntiles = per_cu / (32 * 128);
        for (long long t = wave; t < ntiles; t += nw) {
            char* tb = base + (t / nt_row) * 32 * S + (t % nt_row) * 128;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<int*>(tb + (long long)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * S + (lane & 31) * 4) = lane;
        }
    }
}

template <int FORM>
static void run_store(char* dst, long long per_cu, long long S, int waves, int cus) {
    static const char* names[3] = {"16 B / lane, 1 KiB runs", "4 B / lane, 256 B runs", "4 B / lane, accumulator layout (2 x 128 B runs, rows 4 apart)"};
    printf("  store %-62s %2d waves/CU  row stride %5lld B: ", names[FORM], waves, S);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stream_store<FORM>), dim3(cus), dim3(64 * waves), 0, 0, dst, per_cu, S);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%7.1f GB/s per CU  %6.2f TB/s chip  (%s)\n", (double)per_cu / best * 1e-6, (double)per_cu * cus / best * 1e-9, hipGetErrorString(hipGetLastError()));
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    const long long window = 96 << 10;                      // 32 CUs x 96 KiB = 3 MiB per XCD: L2-resident
    char* src; int* sink;
    hipMalloc(&src, (size_t)cus * (128 << 10) + (4 << 20)); hipMemset(src, 1, (size_t)cus * (128 << 10) + (4 << 20));   // private windows up to 128 KiB; shared ones up to 4 MiB hipMalloc(&sink, 64);
    for (int shared = 0; shared < 2; ++shared) {
        for (int waves : {4, 8, 16}) {
            run<M_DMA16, 8>(src, window, shared, waves, cus, sink);
            run<M_DMA16, 15>(src, window, shared, waves, cus, sink);
            run<M_REG16, 8>(src, window, shared, waves, cus, sink);
            run<M_REG16, 15>(src, window, shared, waves, cus, sink);
            run<M_REG4, 15>(src, window, shared, waves, cus, sink);
            run<M_DMA4, 15>(src, window, shared, waves, cus, sink);
        }
    }
    printf("GEMM-tile fetch pattern (L2-resident):\n");
    for (long long S : {1024LL, 1792LL, 2048LL, 3200LL, 4096LL, 32768LL}) {
        if (64 * S <= (128 << 10)) { run_rows<4>(src, S, false, cus, sink); run_rows<8>(src, S, false, cus, sink); }
        run_rows<4>(src, S, true, cus, sink); run_rows<8>(src, S, true, cus, sink);
    }
    printf("first touch (every byte read once): 1 MiB per CU (256 MiB: Infinity-Cache sized) and 4 MiB per CU (1 GiB: HBM)\n");
    {
        char* big; hipMalloc(&big, (size_t)cus * (4 << 20) + (1 << 20)); hipMemset(big, 1, (size_t)cus * (4 << 20) + (1 << 20));
        for (int waves : {1, 4, 8}) {
            run<M_DMA16, 15>(big, 1 << 20, false, waves, cus, sink, true);
            run<M_DMA16, 15>(big, 4 << 20, false, waves, cus, sink, true);
            run<M_REG16, 15>(big, 4 << 20, false, waves, cus, sink, true);
        }
        printf("stores, 4 MiB per CU written once (1 GiB):\n");
        for (int waves : {4, 8, 16}) {
            run_store<0>(big, 4 << 20, 2048, waves, cus);
            run_store<1>(big, 4 << 20, 2048, waves, cus);
            run_store<2>(big, 4 << 20, 2048, waves, cus);
            run_store<2>(big, 4 << 20, 4096, waves, cus);
        }
        hipFree(big);
    }
    // one CU alone (no contention in L2 / fabric)
    printf("one CU alone:\n");
    for (int waves : {4, 8, 16}) { run<M_DMA16, 15>(src, window, false, waves, 1, sink); run<M_REG16, 15>(src, window, false, waves, 1, sink); }
    // L1-resident window (16 KiB): the vector cache's own rate
    printf("16 KiB window (L1-resident):\n");
    for (int waves : {4, 8}) { run<M_DMA16, 15>(src, 16 << 10, false, waves, cus, sink); run<M_REG16, 15>(src, 16 << 10, false, waves, cus, sink); }
    return 0;
}
