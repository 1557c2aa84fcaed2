// Does a kernel launched with hipExtAnyOrderLaunch (AQL packet without the barrier bit) start beside the kernel queued before it on the SAME stream?
// (the question behind plan option overlap=4: the decode beside the Gram -> miner chain without a cross-stream hop, 15-20 us each on this platform)
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o gpurun_out/anyorder_probe && gpurun_out/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

__global__ void spin(long long ticks, long long* stamp, int slot) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot + 1] = wall_clock64();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    long long* stamp;
    CK(hipMalloc(&stamp, 64 * sizeof(long long)));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks = (long long)rate_khz * 100 / 1000;     // 100 us
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            // A: 128 workgroups (half the chip), in order
            hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, ticks, stamp, 0);
            // B: in order (mode 0), any-order (mode 1), any-order followed by an in-order C (mode 2: C must wait for BOTH)
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, ticks, stamp, 1);
            else hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, stamp, 1);
            if (mode == 2) hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, ticks / 10, stamp, 2);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            long long h[6];
            CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
            const double us = 1e3 / rate_khz;
            printf("mode %d rep %d: %.1f us total | A [0, %.1f]  B [%.1f, %.1f]", mode, rep, ms * 1e3, (h[1] - h[0]) * us, (h[2] - h[0]) * us, (h[3] - h[0]) * us);
            if (mode == 2) printf("  C [%.1f, %.1f]", (h[4] - h[0]) * us, (h[5] - h[0]) * us);
            printf("\n");
        }
    }
    return 0;
}
