// mfma_ubench.hip -- what does one v_mfma_f32_32x32x16_bf16 cost in the instruction streams the dW / dh kernels' consumer waves run?  (gfx950)
// Stand-alone tool (not part of libdae_hip.so):  hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o tools/mfma_ubench
// One workgroup per CU; the MFMA waves (one per SIMD) run ITERS "K tiles" of 4 k steps x 5 MFMAs on 5 accumulators (the 160 x 32 consumer sub-tile of
// gemm_dw_pc) -- or 16 MFMAs on 2 x 2 accumulators (gemm_nt_pc) -- and stamp s_memtime around the loop; printed: shader cycles per MFMA (median over
// the MFMA waves), wall ns per MFMA (HIP events) and the clock the two imply.  Variants add, one at a time, what the real K loop has around the MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { V_BARE = 0, V_BARRIER, V_IDLE_PARTNER, V_DMA_PARTNER, V_READS, V_READS_DMA, V_2X2, V_PAIR10, V_PAIR_READS, V_PAIR_FULL, V_COUNT };
static const char* kNames[V_COUNT] = {
    "5 accumulators, bare MFMAs (4 waves / workgroup)",
    "  + one s_barrier per K tile (20 MFMAs)",
    "  + a second wave per SIMD that only takes part in the barrier",
    "  + that wave issues 9 x 1 KiB global_load_lds per K tile (the producer)",
    "  5 accumulators + 7 ds_read_b128 per k step, counted waits (no partner)",
    "  + producer partner: the complete K loop of gemm_dw_pc",
    "2 x 2 accumulators, 16 MFMAs per K tile (gemm_nt_pc), barrier + idle partner",
    "5 accumulators, 10 MFMAs per k step on alternating B (paired stage), barrier + idle partner",
    "  paired stage + 7 ds_read_b128 per k step (10 MFMAs), barrier + idle partner",
    "  + producer partner issuing 13 x 1 KiB per paired stage (52 KiB, 3-deep ring): the paired K loop"};

__device__ __forceinline__ void mma(const i32x4& a, const i32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ i32x4 lds_read_b128(uint32_t addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=&v"(v) : "v"(addr));
    return v;
}

template <int V>
__global__ __launch_bounds__(512, 1) void probe(long long* out, int iters, const char* src, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool PARTNER = (V == V_IDLE_PARTNER || V == V_DMA_PARTNER || V == V_READS_DMA || V == V_2X2 || V == V_PAIR10 || V == V_PAIR_READS || V == V_PAIR_FULL);
    constexpr bool PAIRV = (V == V_PAIR10 || V == V_PAIR_READS || V == V_PAIR_FULL);
    constexpr bool BARRIER = V != V_BARE && V != V_READS;
    constexpr bool DMA = (V == V_DMA_PARTNER || V == V_READS_DMA || V == V_PAIR_FULL);
    constexpr bool READS = (V == V_READS || V == V_READS_DMA || V == V_PAIR_READS || V == V_PAIR_FULL);
    constexpr int STAGE = (V == V_PAIR_FULL || V == V_PAIR_READS) ? 52 * 1024 : 36 * 1024;
    constexpr int NSLOT = (V == V_PAIR_FULL || V == V_PAIR_READS) ? 3 : 4, NPIECE = V == V_PAIR_FULL ? 13 : 9;
    if (wave >= 4) {                                    // partner wave of the SIMD
        if constexpr (!PARTNER) return;
        const char* g = src + (size_t)blockIdx.x * 4096 + lane * 16;
        for (int it = 0; it < iters; ++it) {
            if constexpr (DMA) {
                char* slot = lds + (it % NSLOT) * STAGE;
#pragma unroll
                for (int i = 0; i < NPIECE; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((it * NPIECE + i) & 63) * 65536),
                                                     (__attribute__((address_space(3))) void*)(slot + (i * 4 + (wave - 4)) * 1024), 16, 0, 0);
                if constexpr (NPIECE == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    f32x16 acc[5];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    i32x4 fa[2][5], fb[2], fb2[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fb[s] = i32x4{(int)(lane * 2654435761u) ^ 0x3f803f80, 0x3e993f12, (int)(lane * 40503u) | 0x3c003c00, 0x3f003e80}; fb2[s] = i32x4{0x3d803f80, (int)(lane * 2246822519u) ^ 0x3e803e80, 0x3f123e99, 0x3e803f00};
#pragma unroll
        for (int m = 0; m < 5; ++m) fa[s][m] = i32x4{(int)((lane + 64 * m) * 3266489917u) ^ 0x3f003f00, 0x3e003f80 + m, (int)(lane * 668265263u) ^ 0x3d003d00, 0x3f803e00};
    }
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    // fragment addressing of the real kernels: row r = lane & 31 of a 128-byte K tile row, 16-byte slot (2 ks + g) XOR-swizzled with (r >> 1) & 7
    const uint32_t rr = lane & 31, gg = lane >> 5, swz = (rr >> 1) & 7;
    uint32_t so[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) so[kk] = rr * 128 + (((kk * 2 + gg) ^ swz) << 4);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint32_t sb = lbase + (it % NSLOT) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int s = ks & 1;
            if constexpr (READS) {                       // the next k step's fragments are requested, the current ones awaited (7 reads per set)
                fb[s ^ 1] = lds_read_b128(sb + 20480 + so[ks]);
                fb2[s ^ 1] = lds_read_b128(sb + 20480 + 8192 + so[ks]);
                fa[s ^ 1][0] = lds_read_b128(sb + so[ks]);
                fa[s ^ 1][1] = lds_read_b128(sb + 4096 + so[ks]);
                fa[s ^ 1][2] = lds_read_b128(sb + 8192 + so[ks]);
                fa[s ^ 1][3] = lds_read_b128(sb + 12288 + so[ks]);
                fa[s ^ 1][4] = lds_read_b128(sb + 16384 + so[ks]);
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else {
                asm volatile("" : "+v"(fb[s]), "+v"(fa[s][0]), "+v"(fa[s][1]), "+v"(fa[s][2]), "+v"(fa[s][3]), "+v"(fa[s][4]));
            }
            if constexpr (V == V_2X2) {                  // 2 x 2 blocks: 4 MFMAs per k step on 4 accumulators
                mma(fa[s][0], fb[s], acc[0]); mma(fa[s][0], fb2[s], acc[1]); mma(fa[s][1], fb[s], acc[2]); mma(fa[s][1], fb2[s], acc[3]);
            } else {
                mma(fa[s][0], fb[s], acc[0]); mma(fa[s][1], fb[s], acc[1]); mma(fa[s][2], fb[s], acc[2]); mma(fa[s][3], fb[s], acc[3]); mma(fa[s][4], fb[s], acc[4]);
                if constexpr (PAIRV) {
                    mma(fa[s][0], fb2[s], acc[0]); mma(fa[s][1], fb2[s], acc[1]); mma(fa[s][2], fb2[s], acc[2]); mma(fa[s][3], fb2[s], acc[3]); mma(fa[s][4], fb2[s], acc[4]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 5; ++m) s += acc[m][lane & 15];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

typedef void (*probe_fn)(long long*, int, const char*, float*);
template <int V> static void fill(probe_fn* t) { t[V] = probe<V>; if constexpr (V + 1 < V_COUNT) fill<V + 1>(t); }
static int mfma_per_iter(int v) { return v == V_2X2 ? 16 : ((v == V_PAIR10 || v == V_PAIR_READS || v == V_PAIR_FULL) ? 40 : 20); }

int main() {
    probe_fn tab[V_COUNT];
    fill<0>(tab);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    long long* d; char* src; float* sink;
    hipMalloc(&d, cus * 4 * sizeof(long long)); hipMalloc(&src, (size_t)64 * 65536 + cus * 4096 + 65536); hipMalloc(&sink, 64);
    hipMemset(src, 0, (size_t)64 * 65536 + cus * 4096 + 65536);
    std::vector<long long> h(cus * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-92s %10s %10s %8s\n", "variant (one workgroup per CU, MFMA waves one per SIMD)", "cyc/MFMA", "ns/MFMA", "GHz");
    for (int v = 0; v < V_COUNT; ++v) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(tab[v]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const bool partner = !(v == V_BARE || v == V_BARRIER || v == V_READS);
        double best_c = 1e30, best_ns = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(tab[v], dim3(cus), dim3(partner ? 512 : 256), 158 * 1024, 0, d, iters, src, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            std::vector<long long> w(h.begin(), h.end());
            std::sort(w.begin(), w.end());
            const double n = (double)iters * mfma_per_iter(v);
            if (rep > 0) { best_c = std::min(best_c, (double)w[w.size() / 2] / n); best_ns = std::min(best_ns, 1e6 * ms / n); }
        }
        printf("%-92s %10.1f %10.2f %8.2f\n", kNames[v], best_c, best_ns, best_c / best_ns);
    }
    return 0;
}
