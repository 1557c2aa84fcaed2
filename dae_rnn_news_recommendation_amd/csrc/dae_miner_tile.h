// dae_miner_tile.h -- the batch_all sweep of one anchor on a 16 x 16 LANE GRID (triplet_loss_utils.py:96-129).
//
// The |P| x |N| rectangle of an anchor is an outer-product-shaped job: every (positive p, negative n) cell needs
// w = 1 + exp(v_n - u_p) = 1 + E_n F_p, its logarithm (loss) and its reciprocal (both gradient roles).  A workgroup of 256
// lanes is laid out as 16 rows x 16 columns: lane (a, b) owns the positives p = a (mod 16) and the negatives
// n = b (mod 16), keeps ITS negatives' E_n and their running column sums in registers for a whole chunk of negatives and walks
// its positives two at a time.  Consequences, against the former layout (one wave = one positive x all negatives):
//   * a positive's row sum is spread over the 16 lanes of ONE DPP row: 4 row-local DPP adds per positive instead of a
//     64-lane reduction chain per positive per wave, and one LDS write per positive;
//   * the negatives' column sums stay in registers until the chunk ends (two xor-shuffles per register, once);
//   * no cell needs a compare: the count of "positive" triplets (T > 1e-16, :114) is taken from a SORT of the anchor's
//     negatives and one binary search per positive (count_positive_triplets below) -- the compare + s_bcnt1 + s_add per cell
//     was ~25 % of the old body's issue time (tools/valu_ubench.hip);
//   * log(w) is only ever needed as a sum: LOGW factors w are multiplied together before one v_log_f32
//     (LOGW = 2 when the row range allows e^80 products only, 4 for range <= 20, 8 for range <= 10).
// Issue cost per pair of cells (MI355X, 3-4 waves per SIMD, tools/valu_ubench.hip): plain VALU ~1.9 cycles, packed fp32
// ~3.3, transcendental ~6.3, v_cmp ~4 + scalar chain.  Old body 44 cycles per pair; this one ~26-30.
//
// Everything is a fixed-order sum: results are deterministic run to run.  Test infrastructure compares it with the oracle
// (tests/test_hip_kernels.py, tests/test_hip_step.py); tools/miner_probe.hip checks and times it stand-alone.
#pragma once
#include "dae_common.h"

namespace dae {

typedef float mt_f32x2 __attribute__((ext_vector_type(2)));
constexpr float kMtLog2e = 1.4426950408889634f;

// ---- single-instruction helpers.  The body below is written instruction by instruction: left to itself hipcc (ROCm 7.2)
// contracts r = swap(w) * R into ONE of its two accumulations and recomputes it for the other, SLP-packs the scalar
// products into v_pk_mul_f32 (packed fp32 costs ~1.7x a plain VALU op here) and keeps every temporary of the unrolled
// body alive (168 VGPRs + scratch inside the loop).  Operand-select forms of the packed ops:
//   op_sel[i]    = which half of source i feeds the LOW result,  op_sel_hi[i] = which half feeds the HIGH result.
// d = a * bcast(b.x) + 1        /  d = a * bcast(b.y) + 1
__device__ __forceinline__ mt_f32x2 mt_w_lo(mt_f32x2 a, mt_f32x2 b) {
    mt_f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
__device__ __forceinline__ mt_f32x2 mt_w_hi(mt_f32x2 a, mt_f32x2 b) {
    mt_f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// d = a * bcast(b.x) + bcast(c.x)   /  d = a * bcast(b.y) + bcast(c.x)      (scaled form: the addend is e^-c instead of 1)
__device__ __forceinline__ mt_f32x2 mt_ws_lo(mt_f32x2 a, mt_f32x2 b, mt_f32x2 c) {
    mt_f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ mt_f32x2 mt_ws_hi(mt_f32x2 a, mt_f32x2 b, mt_f32x2 c) {
    mt_f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
// d = a * bcast(b.x)            /  d = a * bcast(b.y)
__device__ __forceinline__ mt_f32x2 mt_mul_lo(mt_f32x2 a, mt_f32x2 b) {
    mt_f32x2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
__device__ __forceinline__ mt_f32x2 mt_mul_hi(mt_f32x2 a, mt_f32x2 b) {
    mt_f32x2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// acc += swap(w) * bcast(R.x)   /  acc += swap(w) * bcast(R.y)       (swap(w) = {w.y, w.x})
__device__ __forceinline__ void mt_acc_swap_lo(mt_f32x2& acc, mt_f32x2 w, mt_f32x2 R) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "v"(w), "v"(R));
}
__device__ __forceinline__ void mt_acc_swap_hi(mt_f32x2& acc, mt_f32x2 w, mt_f32x2 R) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(R));
}
// d = swap(w) * bcast(R.x)      /  d = swap(w) * bcast(R.y)
__device__ __forceinline__ mt_f32x2 mt_swap_mul_lo(mt_f32x2 w, mt_f32x2 R) {
    mt_f32x2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(d) : "v"(w), "v"(R)); return d;
}
__device__ __forceinline__ mt_f32x2 mt_swap_mul_hi(mt_f32x2 w, mt_f32x2 R) {
    mt_f32x2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(d) : "v"(w), "v"(R)); return d;
}
__device__ __forceinline__ mt_f32x2 mt_pk_add(mt_f32x2 a, mt_f32x2 b) { mt_f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ mt_f32x2 mt_pk_mul(mt_f32x2 a, mt_f32x2 b) { mt_f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ void mt_pk_fma_acc(mt_f32x2& acc, mt_f32x2 a, mt_f32x2 b) { asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
// d = a - 1 (both halves)
__device__ __forceinline__ mt_f32x2 mt_pk_sub1(mt_f32x2 a) { mt_f32x2 d; asm("v_pk_add_f32 %0, %1, -1.0 op_sel_hi:[1,0]" : "=v"(d) : "v"(a)); return d; }
// d = a - b
__device__ __forceinline__ mt_f32x2 mt_pk_sub(mt_f32x2 a, mt_f32x2 b) { mt_f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float mt_mul(float a, float b) { float d; asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float mt_rcp(float a) { float d; asm("v_rcp_f32 %0, %1" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ void mt_add_log2(float& acc, float a) { float l; asm("v_log_f32 %0, %1" : "=v"(l) : "v"(a)); asm("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(l)); }
// acc += log2(a) + k   (scaled form: k = 2c log2(e) restores log2(w_a w_b) >= 0 BEFORE it is accumulated -- summing the raw
// log2(w^_a w^_b) ~ -115 and adding 2c * pairs at the end would cancel five digits)
__device__ __forceinline__ void mt_add_log2k(float& acc, float a, float k) {
    float l; asm("v_log_f32 %0, %1" : "=v"(l) : "v"(a)); asm("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(k)); asm("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(l));
}

// Sum over the 4 DPP rows of a wave for every column b = lane & 15 (lanes l, l^16, l^32, l^48); every lane receives it.
__device__ __forceinline__ float mt_cross_row_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// One chunk of negatives [k0, k0 + 32 Q2) of the compacted list nv[0, nN) against ALL positives pf[0, nP).
//   pf[p]   = F_p = exp(mid - u_p)                      (LDS)
//   nv[n]   = v_n = D[a, n]                              (LDS)
//   gpos[p] : positive-role sums  sum_n sigmoid(v_n - u_p)            (LDS; first chunk stores, later chunks add)
//   gneg_w  : THIS WAVE's partial negative-role sums  sum_{p in the wave's rows} sigmoid(v_n - u_p)   (LDS row [.. nN))
//   loss_log2 += sum log2(w) over this lane's cells;  loss_corr += first-order log1p corrections (natural units; !FAST)
// FAST: accumulates r = 1/w = 1 - sigmoid and converts at the end (padding cells have w = 1, r = 1, sigmoid = 0).
// Per unit of the body (2 positives x 2 negatives, FAST, LOGW = 8): 6 packed + 3.5 plain + 2.5 transcendental instructions.
// SCALED (row range in (40, 80], FAST, LOGW = 2): with c = range / 2 every factor is carried as w^ = e^-c (1 + exp(t)) in
// [e^-c, ~e^c], so that the product of a pair stays finite (e^80) where (1 + e^range)^2 would overflow:
//   1/w_a = e^-c w^_b / (w^_a w^_b),   log w_a + log w_b = log(w^_a w^_b) + 2c
// -- one extra v_mul_f32 (R e^-c) and one extra v_add_f32 (+ 2c) per pair.  `escale` = e^-c (1 when !SCALED).
template <int Q2, bool FAST, int LOGW, int FENCE = 2, bool SCALED = false>
__device__ __forceinline__ void tile_sweep(const float* __restrict__ pf, const float* __restrict__ nv, float mid, int nP, int nN,
                                           int k0, bool first, float* __restrict__ gpos, float* __restrict__ gneg_w,
                                           float& loss_log2, float& loss_corr, int t_begin = 0, int t_step = 1, float escale = 1.0f) {
    static_assert(!SCALED || (FAST && LOGW == 2), "the scaled form exists for the FAST pair sweep with one logarithm per pair");
    // t_begin / t_step: this workgroup walks the positive iterations t_begin, t_begin + t_step, ... (an anchor shared by t_step workgroups)
    static_assert(LOGW == 2 || LOGW == 4 || LOGW == 8, "LOGW");
    const int tid = threadIdx.x, lane = tid & 63, b = tid & 15, a = tid >> 4;
    mt_f32x2 ev2[Q2], gs2[Q2];
#pragma unroll
    for (int q = 0; q < Q2; ++q) {
        const int k = k0 + 32 * q + b;
        const float vx = (k < nN) ? nv[k] : -INFINITY;
        const float vy = (k + 16 < nN) ? nv[k + 16] : -INFINITY;
        ev2[q].x = __builtin_amdgcn_exp2f((vx - mid) * kMtLog2e);          // exp(-inf) = 0 for padding cells
        ev2[q].y = __builtin_amdgcn_exp2f((vy - mid) * kMtLog2e);
        gs2[q] = mt_f32x2{0.f, 0.f};
    }
    mt_f32x2 corr2 = {0.f, 0.f};
    const mt_f32x2 esc2 = {escale, escale};
    const float klog = SCALED ? -2.0f * __builtin_amdgcn_logf(escale) : 0.f;       // 2c log2(e) = -2 log2(e^-c)
    (void)esc2; (void)klog;
    const int T = (nP + 31) >> 5;                                           // two positives per iteration: a + 32 t, a + 32 t + 16
    mt_f32x2 Fn;
    Fn.x = (a + 32 * t_begin < nP) ? pf[a + 32 * t_begin] : 0.f; Fn.y = (a + 32 * t_begin + 16 < nP) ? pf[a + 32 * t_begin + 16] : 0.f;
    int walked = 0;
    for (int t = t_begin; t < T; t += t_step) {
        ++walked;
        const int j0 = a + 32 * t, j1 = j0 + 16;
        mt_f32x2 ff = Fn;
        if constexpr (SCALED) { ff.x *= escale; ff.y *= escale; }
        Fn.x = (j0 + 32 * t_step < nP) ? pf[j0 + 32 * t_step] : 0.f;        // next iteration's factors (LDS latency under the body)
        Fn.y = (j1 + 32 * t_step < nP) ? pf[j1 + 32 * t_step] : 0.f;
        mt_f32x2 rs0 = {0.f, 0.f}, rs1 = {0.f, 0.f};
        float PP = 1.0f;
#pragma unroll
        for (int q = 0; q < Q2; ++q) {
            float P0, P1;
            mt_f32x2 RR;
            if constexpr (FAST) {
                mt_f32x2 w0, w1;
                if constexpr (SCALED) { w0 = mt_ws_lo(ev2[q], ff, esc2); w1 = mt_ws_hi(ev2[q], ff, esc2); }   // e^-c (1 + exp(t)); ff carries e^-c too
                else { w0 = mt_w_lo(ev2[q], ff); w1 = mt_w_hi(ev2[q], ff); }                                   // 1 + exp(t)
                P0 = mt_mul(w0.x, w0.y); P1 = mt_mul(w1.x, w1.y);
                RR.x = mt_rcp(P0); RR.y = mt_rcp(P1);
                if constexpr (SCALED) { RR.x = mt_mul(RR.x, escale); RR.y = mt_mul(RR.y, escale); }
                mt_acc_swap_lo(gs2[q], w0, RR); mt_acc_swap_hi(gs2[q], w1, RR);           // += 1/w = 1 - sigmoid(t)
                mt_acc_swap_lo(rs0, w0, RR); mt_acc_swap_hi(rs1, w1, RR);
            } else {
                const mt_f32x2 e0 = mt_mul_lo(ev2[q], ff), e1 = mt_mul_hi(ev2[q], ff);     // exp(t)
                const mt_f32x2 one = {1.0f, 1.0f};
                const mt_f32x2 w0 = mt_pk_add(e0, one), w1 = mt_pk_add(e1, one);
                P0 = mt_mul(w0.x, w0.y); P1 = mt_mul(w1.x, w1.y);
                RR.x = mt_rcp(P0); RR.y = mt_rcp(P1);
                const mt_f32x2 r0 = mt_swap_mul_lo(w0, RR), r1 = mt_swap_mul_hi(w1, RR);   // 1/w
                // log1p(e) = log(fl(1+e)) + (e - (fl(1+e) - 1)) / fl(1+e)
                mt_pk_fma_acc(corr2, mt_pk_sub(e0, mt_pk_sub1(w0)), r0);
                mt_pk_fma_acc(corr2, mt_pk_sub(e1, mt_pk_sub1(w1)), r1);
                mt_pk_fma_acc(gs2[q], e0, r0); mt_pk_fma_acc(gs2[q], e1, r1);             // += sigmoid(t) = e / w
                mt_pk_fma_acc(rs0, e0, r0); mt_pk_fma_acc(rs1, e1, r1);
            }
            // sum of logs = log of the product: LOGW factors w per v_log_f32
            if constexpr (LOGW == 2) {
                if constexpr (SCALED) { mt_add_log2k(loss_log2, P0, klog); mt_add_log2k(loss_log2, P1, klog); }
                else { mt_add_log2(loss_log2, P0); mt_add_log2(loss_log2, P1); }
            } else if constexpr (LOGW == 4) {
                mt_add_log2(loss_log2, mt_mul(P0, P1));
            } else {
                if ((q & 1) == 0 && q + 1 < Q2) PP = mt_mul(P0, P1);
                else mt_add_log2(loss_log2, (q & 1) ? mt_mul(PP, mt_mul(P0, P1)) : mt_mul(P0, P1));
            }
            // fence for the machine scheduler every FENCE register pairs: FENCE x two positives independent chains in flight, no more
            if (FENCE > 0 && (q % FENCE) == FENCE - 1) __builtin_amdgcn_sched_barrier(0);
        }
        float s0 = row16_sum(rs0.x + rs0.y), s1 = row16_sum(rs1.x + rs1.y);
        if constexpr (FAST) { s0 = (float)(32 * Q2) - s0; s1 = (float)(32 * Q2) - s1; }      // sum of sigmoids = cells - sum of r
        if (b == 0) {
            if (first) { if (j0 < nP) gpos[j0] = s0; if (j1 < nP) gpos[j1] = s1; }
            else { if (j0 < nP) gpos[j0] += s0; if (j1 < nP) gpos[j1] += s1; }
        }
    }
    loss_corr += corr2.x + corr2.y;
    const float slots = (float)(8 * walked);                                // positive cells walked by the 4 rows of this wave
#pragma unroll
    for (int q = 0; q < Q2; ++q) {
        const float cx = mt_cross_row_sum(gs2[q].x), cy = mt_cross_row_sum(gs2[q].y);
        const int k = k0 + 32 * q + b;
        if (lane < 16) {
            if (k < nN) gneg_w[k] = FAST ? slots - cx : cx;
            if (k + 16 < nN) gneg_w[k + 16] = FAST ? slots - cy : cy;
        }
    }
}

// Bitonic sort of 256 floats inside ONE wave, ascending over the index e = 4 * lane + r (x[r] of lane `lane`): the strides 1
// and 2 of the network are register-to-register, the strides 4 .. 128 one cross-lane exchange per register -- no LDS
// array, no barrier (an LDS-resident network cost 42 k cycles per anchor, 763 per stage, against 29 k for the whole sweep).
// A compare-exchange output is ONE v_med3_f32: med3(x, y, -inf) = min(x, y), med3(x, y, +inf) = max(x, y), so the direction
// is a per-lane constant instead of min + max + select.  Lane distances 1, 2, 4, 8 are DPP moves on the VALU operand path
// (quad_perm, row_shl / row_shr under bank masks, row_ror:8); only the distances 16 and 32 (3 of the 21 cross-lane stages) go
// through the LDS crossbar.  No NaNs reach this (NaN rows never take the count's fast path ... they give NaN statistics anyway).
__device__ __forceinline__ float mt_med3(float a, float b, float c) { float d; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
template <int DIST>
__device__ __forceinline__ float mt_lane_xor(float f) {
    const int v = __float_as_int(f);
    int r;
    if constexpr (DIST == 1) r = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);               // quad_perm [1,0,3,2]
    else if constexpr (DIST == 2) r = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);          // quad_perm [2,3,0,1]
    else if constexpr (DIST == 4) {
        r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);                                   // banks 0, 2 <- lane + 4   (row_shl:4)
        r = __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);                                   // banks 1, 3 <- lane - 4   (row_shr:4)
    } else if constexpr (DIST == 8) r = __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);       // row_ror:8
    else r = __shfl_xor(v, DIST, 64);
    return __int_as_float(r);
}
__device__ __forceinline__ void mt_cmpx(float& lo, float& hi, float c_lo, float c_hi) {   // c_lo = -inf, c_hi = +inf: (lo, hi) ascending
    const float a = mt_med3(lo, hi, c_lo), b = mt_med3(lo, hi, c_hi);
    lo = a; hi = b;
}
template <int K, int J>
__device__ __forceinline__ void mt_cross_stage(float (&x)[4], int lane) {       // partner element e ^ J lives in lane ^ (J / 4), same register
    const bool asc = K == 256 ? true : ((lane & (K >> 2)) == 0);
    const bool lower = (lane & (J >> 2)) == 0;
    const float c = (lower == asc) ? -INFINITY : INFINITY;                      // keep the smaller / the larger of the two
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = mt_med3(x[r], mt_lane_xor<(J >> 2)>(x[r]), c);
}
template <int K>
__device__ __forceinline__ void mt_sort_level(float (&x)[4], int lane) {
    // blocks of K elements alternate ascending / descending ((e & K) == 0 <=> ascending; the last merge is ascending)
    if constexpr (K >= 256) mt_cross_stage<K, 128>(x, lane);
    if constexpr (K >= 128) mt_cross_stage<K, 64>(x, lane);
    if constexpr (K >= 64) mt_cross_stage<K, 32>(x, lane);
    if constexpr (K >= 32) mt_cross_stage<K, 16>(x, lane);
    if constexpr (K >= 16) mt_cross_stage<K, 8>(x, lane);
    if constexpr (K >= 8) mt_cross_stage<K, 4>(x, lane);
    if constexpr (K >= 4) {
        const bool asc = K == 256 ? true : ((lane & (K >> 2)) == 0);           // K = 4: (e & 4) = lane & 1
        const float c_lo = asc ? -INFINITY : INFINITY, c_hi = asc ? INFINITY : -INFINITY;
        mt_cmpx(x[0], x[2], c_lo, c_hi); mt_cmpx(x[1], x[3], c_lo, c_hi);       // stride 2
        mt_cmpx(x[0], x[1], c_lo, c_hi); mt_cmpx(x[2], x[3], c_lo, c_hi);       // stride 1
    } else {                                                                    // K = 2: pairs (0,1) ascending, (2,3) descending
        mt_cmpx(x[0], x[1], -INFINITY, INFINITY); mt_cmpx(x[2], x[3], INFINITY, -INFINITY);
    }
}
__device__ __forceinline__ void wave_sort256(float (&x)[4], int lane) {
    mt_sort_level<2>(x, lane); mt_sort_level<4>(x, lane); mt_sort_level<8>(x, lane); mt_sort_level<16>(x, lane);
    mt_sort_level<32>(x, lane); mt_sort_level<64>(x, lane); mt_sort_level<128>(x, lane); mt_sort_level<256>(x, lane);
}

// Number of "positive" triplets of one anchor, #{(p, n) : fl(v_n - u_p) > 1e-16f} (triplet_loss_utils.py:106,114), exact:
// fl(v - u) is monotone in v, so once negatives are sorted ascending the cells that satisfy the reference's literal predicate
// form a suffix for every positive, and its start is found by binary search WITH that predicate.  Each of the 4 waves sorts
// a quarter of the negatives (256 slots, padding +inf) in registers and parks its run in `runs` (LDS, 1024 floats, nothing
// live in it); every positive is then searched in the four runs at once (8 rounds of 4 independent LDS reads).
// Returns this thread's share (the block sum is the anchor's count).  nN <= 1024, blockDim.x = 256.
__device__ __forceinline__ unsigned count_positive_triplets(const float* __restrict__ pu, int nP, const float* __restrict__ nv, int nN,
                                                            float* __restrict__ runs) {
    const int tid = threadIdx.x, lane = tid & 63;
    float x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int i = 4 * tid + r; x[r] = (i < nN) ? nv[i] : INFINITY; }
    wave_sort256(x, lane);
    *reinterpret_cast<float4*>(runs + 4 * tid) = float4{x[0], x[1], x[2], x[3]};
    __syncthreads();
    unsigned cnt = 0u;
    const int pad = 1024 - nN;                                              // the +inf slots pass the predicate for every positive
    for (int p = tid; p < nP; p += 256) {
        const float u = pu[p];
        int i0 = 0, i1 = 0, i2 = 0, i3 = 0;                                 // per run: number of slots that FAIL the predicate (a prefix)
#pragma unroll
        for (int s = 128; s > 0; s >>= 1) {
            const float a0 = runs[i0 + s - 1], a1 = runs[256 + i1 + s - 1], a2 = runs[512 + i2 + s - 1], a3 = runs[768 + i3 + s - 1];
            if (!(a0 - u > 1e-16f)) i0 += s;
            if (!(a1 - u > 1e-16f)) i1 += s;
            if (!(a2 - u > 1e-16f)) i2 += s;
            if (!(a3 - u > 1e-16f)) i3 += s;
        }
        // the search above finds up to 255 failing slots per run; the last slot is checked on its own
        i0 += !(runs[i0] - u > 1e-16f) && i0 == 255 ? 1 : 0;
        i1 += !(runs[256 + i1] - u > 1e-16f) && i1 == 255 ? 1 : 0;
        i2 += !(runs[512 + i2] - u > 1e-16f) && i2 == 255 ? 1 : 0;
        i3 += !(runs[768 + i3] - u > 1e-16f) && i3 == 255 ? 1 : 0;
        cnt += (unsigned)(1024 - (i0 + i1 + i2 + i3) - pad);
    }
    return cnt;
}

}  // namespace dae
