// dae_common.h -- shared device helpers for the gfx950 DAE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
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

// ---- kernel launches ----
// Every kernel of the library is launched through DAE_LAUNCH.  Normally that is hipLaunchKernelGGL.  While dae_train_step runs in profile
// mode 3 (dae_plan_profile, "kernel timestamps") the launch takes an event pair from the plan's pool and goes through hipExtLaunchKernelGGL,
// which stamps the pair with the dispatch's OWN begin / end times -- what rocprofv3 --kernel-trace reports -- instead of the times of marker
// packets around it; nothing is added to the stream and the host never waits between launches.
struct LaunchTimer {
    hipEvent_t* pool; int* used; int* slots; int cap;   // the plan's event pool (pairs), pairs handed out so far, slot of every pair
    int slot; bool first;                                 // slot of the PROF call being executed; its first launch carries the call count
};
extern thread_local LaunchTimer g_lt;
inline bool launch_timer_take(hipEvent_t& e0, hipEvent_t& e1) {
    LaunchTimer& t = g_lt;
    if (!t.pool || *t.used + 2 > t.cap) return false;
    e0 = t.pool[*t.used]; e1 = t.pool[*t.used + 1];
    t.slots[*t.used / 2] = t.slot | (t.first ? 0x100 : 0); t.first = false; *t.used += 2;
    return true;
}
#define DAE_LAUNCH(kernel, grid, block, lds, st, ...)                                                              \
    do {                                                                                                           \
        hipEvent_t lt0__, lt1__;                                                                                   \
        if (dae::launch_timer_take(lt0__, lt1__)) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, lt0__, lt1__, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                        \
    } while (0)

// ---- the library's 16-bit storage format ----
// Every 16-bit operand image (W shadows, h, x~^T, delta2, delta1, Gs, Gram operands) is written and read through the helpers below, and the
// MFMA that multiplies them is chosen from the same switch (dae_gemm.hip: Mma<bf16_t>).  The library is built twice from the same sources:
//   libdae_hip.so      DAE_F16 = 0   bfloat16 (8 significant bits, fp32 range)          v_mfma_f32_32x32x16_bf16
//   libdae_hip_f16.so  DAE_F16 = 1   IEEE fp16 (11 significant bits, normal range 6e-5 .. 65504)   v_mfma_f32_32x32x16_f16
// Both MFMAs run at the same rate; fp16's three extra bits are what lets the parity mode multiply TWO product terms per gradient GEMM instead of
// split-bf16's three (tools/precision_study.py --scheme).  fp16's narrow range is handled by the plan: the back-propagated images
// (delta2, Gs, delta1) are stored times a power of two `op_scale` that the consuming epilogues divide out again (dae_api.hip).
// The type name bf16_t ("raw 16 bits") and the f2bf* / bf2f names are kept for both formats.
#ifndef DAE_F16
#define DAE_F16 0
#endif
constexpr bool kF16 = DAE_F16 != 0;
constexpr uint32_t kOne16 = kF16 ? 0x3C00u : 0x3F80u;       // 1.0 in the storage format
// round to nearest even, NaN preserved (fp16: overflow -> inf, subnormals kept)
__device__ __forceinline__ bf16_t f2bf(float f) {
    if constexpr (kF16) return __builtin_bit_cast(bf16_t, static_cast<_Float16>(f));
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
    if constexpr (kF16) return static_cast<float>(__builtin_bit_cast(_Float16, b));
    return __uint_as_float(((uint32_t)b) << 16);
}
// hardware RNE convert (v_cvt_pk_bf16_f32 / v_cvt_f16_f32 on gfx950); identical to f2bf for finite inputs
__device__ __forceinline__ bf16_t f2bf_hw(float f) {
    if constexpr (kF16) return __builtin_bit_cast(bf16_t, static_cast<_Float16>(f));
    return __builtin_bit_cast(bf16_t, static_cast<__bf16>(f));
}
__device__ __forceinline__ uint32_t f2bf_pack_hw(float lo, float hi) {
    if constexpr (kF16) {
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        f16x2_t v = {static_cast<_Float16>(lo), static_cast<_Float16>(hi)};
        return __builtin_bit_cast(uint32_t, v);
    }
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};
    return __builtin_bit_cast(uint32_t, v);
}
// a scaled back-propagated value on its way into a 16-bit image: fp16 saturates at +-65504 instead of overflowing to inf (an inf would turn
// every product with a zero operand into NaN); bf16 has the fp32 range and needs nothing
__device__ __forceinline__ float sat16(float v) {
    if constexpr (kF16) return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
    return v;
}
// host side: the 16-bit image of a finite float (round to nearest even) -- operands the launchers build themselves (the value of a kept entry)
static inline uint32_t host_f2bf(float f) {
    if (kF16) { const _Float16 h = static_cast<_Float16>(f); uint16_t b; memcpy(&b, &h, 2); return b; }
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float from(float v) { return v; }
    static __device__ __forceinline__ float to(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ bf16_t from(float v) { return f2bf(v); }
    static __device__ __forceinline__ float to(bf16_t v) { return bf2f(v); }
};

// split-bf16 operands (precision mode bf16x3): x = hi + lo with hi = bf16(x) and lo = bf16(x - hi); products multiply (hi,hi) + (hi,lo) + (lo,hi)
template <typename T> __device__ __forceinline__ T elem_residual(float v) { return Elem<T>::from(v - Elem<T>::to(Elem<T>::from(v))); }
__device__ __forceinline__ bf16_t bf_residual_hw(float v) { return f2bf_hw(v - bf2f(f2bf_hw(v))); }
__device__ __forceinline__ uint32_t bf_residual_pack_hw(float a, float b) {
    return f2bf_pack_hw(a - bf2f(f2bf_hw(a)), b - bf2f(f2bf_hw(b)));
}

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

// ---- DPP cross-lane sums (no LDS traffic): gfx9 data-parallel primitives on the VALU operand path ----
// dpp_ctrl: quad_perm = 8-bit lane permutation; 0x140 row_mirror; 0x141 row_half_mirror;
//           0x142 row_bcast15 (lane 15 of each 16-lane row -> next row); 0x143 row_bcast31 (lane 31 -> rows 2,3)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xF, BOUND));
}
// sum over each 16-lane row; every lane of the row receives it
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1, 0xF, true>(0.f, v);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xF, true>(0.f, v);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xF, true>(0.f, v);    // row_half_mirror: the other quad of each 8
    v += dpp_mov<0x140, 0xF, true>(0.f, v);    // row_mirror: the other 8 of each 16
    return v;
}
// sum over each 32-lane half wave; valid in lanes 16..31 and 48..63
__device__ __forceinline__ float half32_sum_hi(float v) {
    v = row16_sum(v);
    return v + dpp_mov<0x142, 0xA, false>(0.f, v);   // rows 1 and 3 += lane 15 of rows 0 and 2
}
// sum over all 64 lanes; valid in lanes 48..63
__device__ __forceinline__ float wave64_sum_hi(float v) {
    v = half32_sum_hi(v);
    return v + dpp_mov<0x143, 0xC, false>(0.f, v);   // rows 2,3 += lane 31
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
