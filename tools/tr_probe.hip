// tr_probe.hip -- pins the lane <-> element mapping of gfx950's transposing LDS read (ds_read_b64_tr_b16) on the box.
// LDS holds a [64 k][PITCH/2 m] 16-bit matrix, element (k, m) = k * 256 + m.  Every lane supplies the address of 4 consecutive m of ONE k row;
// the instruction returns, per lane, 4 elements.  Printed: for each lane the (k, m) of the 4 values it received, for the addressing this repo
// would use to read an MFMA 32x32x16 A fragment (rows = m, 8 consecutive k per lane) out of a [k][m] image:
//     group q = lane >> 4, i = lane & 15:  row k0 + (i >> 2), columns m0 + 4 * (i & 3) .. + 3,  with m0 = 16 * (q & 1), k0 = 8 * (q >> 1) (+ 4 for the second read)
// build: hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 320;   // bytes per k row (160 m)
__global__ void probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * PITCH / 2];
    for (int e = threadIdx.x; e < 64 * PITCH / 2; e += 64) lds[e] = (uint16_t)((e / (PITCH / 2)) * 256 + e % (PITCH / 2));
    __syncthreads();
    const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
    const int m0 = 16 * (q & 1), k0 = 8 * (q >> 1);
    for (int t = 0; t < 2; ++t) {
        const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + (uint32_t)((k0 + 4 * t + (i >> 2)) * PITCH + (m0 + 4 * (i & 3)) * 2);
        s16x4 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        for (int j = 0; j < 4; ++j) out[(lane * 2 + t) * 4 + j] = (uint16_t)v[j];
    }
}
int main() {
    uint16_t* d; uint16_t h[64 * 8];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int lane = 0; lane < 64; ++lane) {
        const int q = lane >> 4, i = lane & 15, m = 16 * (q & 1) + i, k0 = 8 * (q >> 1);
        printf("lane %2d (wants m=%2d k=%d..%d):", lane, m, k0, k0 + 7);
        for (int e = 0; e < 8; ++e) {
            const int k = h[lane * 8 + e] >> 8, mm = h[lane * 8 + e] & 255;
            printf(" (%d,%d)", k, mm);
            if (k != k0 + e || mm != m) ok = 0;
        }
        printf("\n");
    }
    printf("A-fragment addressing %s\n", ok ? "CONFIRMED: lane gets A[m][k0..k0+7]" : "DIFFERS from the assumption");
    return 0;
}
