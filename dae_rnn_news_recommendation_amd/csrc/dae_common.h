// dae_common.h -- shared device helpers for the gfx950 DAE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dae_hip.h"

namespace dae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;   // raw bf16 bits

constexpr int WAVE = 64;

// ---- error plumbing (host) ----
void set_error(const char* fmt, ...);
#define DAE_CHECK_ARG(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) { dae::set_error(__VA_ARGS__); return 1; }    \
    } while (0)
#define DAE_CHECK_HIP(expr)                                                                    \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            dae::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return 2;                                                                          \
        }                                                                                      \
    } while (0)
#define DAE_CHECK_LAUNCH() DAE_CHECK_HIP(hipGetLastError())

// ---- bf16 <-> f32 (round to nearest even, NaN preserved) ----
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float from(float v) { return v; }
    static __device__ __forceinline__ float to(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ bf16_t from(float v) { return f2bf(v); }
    static __device__ __forceinline__ float to(bf16_t v) { return bf2f(v); }
};

// ---- activations (autoencoder.py:380-389, 398-411) ----
__device__ __forceinline__ float sigmoidf_(float z) {
    // stable logistic; __expf error ~2 ulp, far inside the 1e-4 loss gate
    float e = __expf(-fabsf(z));
    float r = 1.0f / (1.0f + e);
    return z >= 0.f ? r : e * r;
}
__device__ __forceinline__ float act_apply(int act, float z) {
    if (act == DAE_ACT_SIGMOID) return sigmoidf_(z);
    if (act == DAE_ACT_TANH) return tanhf(z);
    return z;
}
// derivative written in terms of the activation OUTPUT a (TF SigmoidGrad / TanhGrad)
__device__ __forceinline__ float act_grad(int act, float a) {
    if (act == DAE_ACT_SIGMOID) return a * (1.0f - a);
    if (act == DAE_ACT_TANH) return 1.0f - a * a;
    return 1.0f;
}

// tf.nn.softplus with TF 1.12's thresholds (threshold = log(eps_f32) + 2 = -13.9424)
__device__ __forceinline__ float softplus_tf(float x) {
    const float thr = -13.942384f;
    if (x > -thr) return x;
    float e = __expf(x);
    if (x < thr) return e;
    return log1pf(e);
}

// ---- wave reductions over 64 lanes (wavefront shuffles) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int64_t pad128(int64_t n) { return (n + DAE_PAD - 1) / DAE_PAD * DAE_PAD; }

}  // namespace dae
