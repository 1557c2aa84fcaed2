// What does a cross-stream dependency cost on this platform, by mechanism?  Kernel A (stream 1) -> kernel B (stream 2):
//   E  hipEventRecord + hipStreamWaitEvent                        (what plan option overlap = 1..3 uses)
//   W  hipStreamWriteValue32 behind A + hipStreamWaitValue32       (command-processor memory ops on signal memory)
//   V  A's last workgroup stores the flag itself + hipStreamWaitValue32 on stream 2
//   S  B is launched at once and its workgroups spin on a flag in device memory that A's last workgroup stores
// Printed: B's first timestamp minus A's last (device wall clock), and the total by events.
//   hipcc --offload-arch=gfx950 -O2 tools/hop_probe.hip -o /tmp/hop_probe && /tmp/hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void work(long long ticks, long long* stamp, int slot, unsigned* counter, unsigned* flag, unsigned value, const unsigned* spin_on, unsigned spin_value) {
    if (spin_on) {
        if (threadIdx.x == 0) atomicMin((unsigned long long*)&stamp[4], (unsigned long long)wall_clock64());       // when B's workgroups arrived
        if (threadIdx.x == 0) while (__hip_atomic_load(spin_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < spin_value) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
    }
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin((unsigned long long*)&stamp[2 * slot], (unsigned long long)t0);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax((unsigned long long*)&stamp[2 * slot + 1], (unsigned long long)wall_clock64());
        if (counter) {
            __threadfence();
            if (atomicAdd(counter, 1u) == gridDim.x - 1 && flag) {
                *counter = 0;
                __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    long long* stamp; unsigned *counter, *dflag, *sig;
    CK(hipMalloc(&stamp, 64 * sizeof(long long)));
    CK(hipMalloc(&counter, 64)); CK(hipMemset(counter, 0, 64));
    CK(hipExtMallocWithFlags((void**)&dflag, 64, (getenv("HOP_FINE") ? hipDeviceMallocFinegrained : hipDeviceMallocUncached))); CK(hipMemset(dflag, 0, 64));
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); CK(hipMemset(sig, 0, 8));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks = (long long)rate_khz * 40 / 1000;     // 40 us
    hipEvent_t e0, e1, ev;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const char* names = "EWVSC";
    unsigned gen = 0;
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 5; ++rep) {
            ++gen;
            long long init[5] = {0x7fffffffffffffffLL, 0, 0x7fffffffffffffffLL, 0, 0x7fffffffffffffffLL};
            CK(hipMemcpy(stamp, init, sizeof(init), hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s1));
            if (mode == 0) {
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, ticks, stamp, 0, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (const unsigned*)nullptr, 0u);
                CK(hipEventRecord(ev, s1));
                CK(hipStreamWaitEvent(s2, ev, 0));
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s2, ticks / 4, stamp, 1, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (const unsigned*)nullptr, 0u);
            } else if (mode == 1) {
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, ticks, stamp, 0, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (const unsigned*)nullptr, 0u);
                CK(hipStreamWriteValue32(s1, sig, gen, 0));
                CK(hipStreamWaitValue32(s2, sig, gen, hipStreamWaitValueGte, 0xffffffffu));
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s2, ticks / 4, stamp, 1, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (const unsigned*)nullptr, 0u);
            } else if (mode == 2) {
                CK(hipStreamWaitValue32(s2, sig, gen, hipStreamWaitValueGte, 0xffffffffu));
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s2, ticks / 4, stamp, 1, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (const unsigned*)nullptr, 0u);
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, ticks, stamp, 0, counter, sig, gen, (const unsigned*)nullptr, 0u);
            } else if (mode == 4) {     // no dependency at all: when does B start beside A?
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, ticks, stamp, 0, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (const unsigned*)nullptr, 0u);
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s2, ticks / 4, stamp, 1, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (const unsigned*)nullptr, 0u);
            } else {
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, ticks, stamp, 0, counter, dflag, gen, (const unsigned*)nullptr, 0u);
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s2, ticks / 4, stamp, 1, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (const unsigned*)dflag, gen);
            }
            // join back on stream 1 by event (not what is measured)
            CK(hipEventRecord(ev, s2));
            CK(hipStreamWaitEvent(s1, ev, 0));
            CK(hipEventRecord(e1, s1));
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            long long h[5];
            CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
            const double us = 1e3 / rate_khz;
            printf("%c rep %d: A [0, %.1f]  B [%.1f, %.1f]  hop %.1f us   total %.1f us (incl. an event join)\n", names[mode], rep, (h[1] - h[0]) * us, (h[2] - h[0]) * us,
                   (h[3] - h[0]) * us, (h[2] - h[1]) * us, ms * 1e3);
            if (mode == 3) printf("      B's workgroups arrived at %.1f\n", (h[4] - h[0]) * us);
        }
    }
    return 0;
}
