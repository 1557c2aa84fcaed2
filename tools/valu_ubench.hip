// valu_ubench.hip -- issue-rate probe for the VALU instructions the triplet miner's sweep is built from (gfx950).
// Stand-alone tool (not part of libdae_hip.so):  hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o tools/valu_ubench
// For every instruction kind: one workgroup per CU with W waves per SIMD (W = 1..4), each wave runs N back-to-back
// instructions on 8 independent register chains and stamps s_memtime around them; printed: shader cycles per instruction
// per WAVE and per SIMD (= per wave / W).  Composite bodies at the end are candidate inner loops of the miner.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { K_FMA = 0, K_PKFMA, K_PKADD, K_PKMUL, K_MUL, K_ADD, K_RCP, K_LOG, K_EXP, K_CMP_BCNT, K_CMP_ONLY, K_DPP_ADD, K_BODY_PK, K_BODY_SC,
            K_BODY_NEW, K_BODY_NEW4, K_BODY_NEWSC, K_PKFMA_BCAST, K_PKFMA_SWAP, K_PKFMA_CONST, K_UNIT_PK, K_UNIT_SC, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_mul_f32", "v_add_f32", "v_rcp_f32",
                                      "v_log_f32", "v_exp_f32", "v_cmp+s_bcnt1+s_add", "v_cmp only", "v_add_f32 dpp row_shr",
                                      "pair body: current (packed ops, log+rcp per pair, t2+2cmp)",
                                      "pair body: same arithmetic, scalar ops",
                                      "pair body: new packed (no t2, log per 2 pairs)",
                                      "pair body: new packed (no t2, log per 4 pairs)",
                                      "pair body: new scalar (no t2, log per 2 pairs)",
                                      "v_pk_fma_f32 op_sel_hi:[1,0,1] (src1 low half broadcast)",
                                      "v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,0,1] (src0 swapped, src1 broadcast)",
                                      "v_pk_fma_f32 a, b, 1.0 op_sel_hi:[1,0,0] (inline constant)",
                                      "lane-grid unit (2 pos x 2 neg): packed, as tile_sweep",
                                      "lane-grid unit (2 pos x 2 neg): scalar v_fma_f32 only"};
// instructions (or pairs, for the bodies) per loop iteration
static const int kPerIter[K_COUNT] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 8, 8, 8, 8, 8, 32, 32, 32, 8, 8};

template <int KIND>
__global__ __launch_bounds__(1024) void probe(long long* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float a[8], b = seed + 1e-3f * lane, c = 1.0f + 1e-4f * lane;
    f32x2 p[8], q = {b, c}, r = {c, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + lane; p[i] = f32x2{a[i], a[i] + 1.f}; }
    unsigned cnt = 0;
    float L = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q), "v"(r));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(r));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_LOG) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_CMP_BCNT) {
#define X(i) { unsigned n; asm volatile("v_cmp_gt_f32 vcc, %1, %2\n s_bcnt1_i32_b64 %0, vcc" : "=s"(n) : "v"(a[i]), "v"(b) : "vcc", "scc"); cnt += n; }
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_CMP_ONLY) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_DPP_ADD) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_PKFMA_BCAST) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(q), "v"(r));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_PKFMA_SWAP) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(p[i]) : "v"(q), "v"(r));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_PKFMA_CONST) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel_hi:[1,0,0]" : "=v"(p[i]) : "v"(q), "v"(r));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == K_UNIT_PK) {
            // 8 units: ev2 = p[i], factors ff = q, accumulators: gs = p[i] (stand-in: separate regs below), rs0/rs1
            f32x2 rs0 = {0.f, 0.f}, rs1 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x2 w0, w1, RR;
                asm("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel_hi:[1,0,0]" : "=v"(w0) : "v"(p[i]), "v"(q));
                asm("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(w1) : "v"(p[i]), "v"(q));
                float P0, P1, PP, lg;
                asm("v_mul_f32 %0, %1, %2" : "=v"(P0) : "v"(w0.x), "v"(w0.y));
                asm("v_mul_f32 %0, %1, %2" : "=v"(P1) : "v"(w1.x), "v"(w1.y));
                asm("v_rcp_f32 %0, %1" : "=v"(RR.x) : "v"(P0));
                asm("v_rcp_f32 %0, %1" : "=v"(RR.y) : "v"(P1));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(w0), "v"(RR));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(r) : "v"(w1), "v"(RR));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(rs0) : "v"(w0), "v"(RR));
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(rs1) : "v"(w1), "v"(RR));
                asm("v_mul_f32 %0, %1, %2" : "=v"(PP) : "v"(P0), "v"(P1));
                asm("v_log_f32 %0, %1" : "=v"(lg) : "v"(PP));
                asm("v_add_f32 %0, %0, %1" : "+v"(L) : "v"(lg));
            }
            L += rs0.x + rs0.y + rs1.x + rs1.y;
        } else if constexpr (KIND == K_UNIT_SC) {
            float s0 = 0.f, s1 = 0.f, gx = 0.f, gy = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float w0x, w0y, w1x, w1y, P0, P1, R0, R1, PP, lg;
                asm("v_fma_f32 %0, %1, %2, 1.0" : "=v"(w0x) : "v"(p[i].x), "v"(q.x));
                asm("v_fma_f32 %0, %1, %2, 1.0" : "=v"(w0y) : "v"(p[i].y), "v"(q.x));
                asm("v_fma_f32 %0, %1, %2, 1.0" : "=v"(w1x) : "v"(p[i].x), "v"(q.y));
                asm("v_fma_f32 %0, %1, %2, 1.0" : "=v"(w1y) : "v"(p[i].y), "v"(q.y));
                asm("v_mul_f32 %0, %1, %2" : "=v"(P0) : "v"(w0x), "v"(w0y));
                asm("v_mul_f32 %0, %1, %2" : "=v"(P1) : "v"(w1x), "v"(w1y));
                asm("v_rcp_f32 %0, %1" : "=v"(R0) : "v"(P0));
                asm("v_rcp_f32 %0, %1" : "=v"(R1) : "v"(P1));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(gx) : "v"(w0y), "v"(R0));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(gy) : "v"(w0x), "v"(R0));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(gx) : "v"(w1y), "v"(R1));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(gy) : "v"(w1x), "v"(R1));
                float t0, t1;
                asm("v_add_f32 %0, %1, %2" : "=v"(t0) : "v"(w0x), "v"(w0y));
                asm("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(w1x), "v"(w1y));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(t0), "v"(R0));
                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(t1), "v"(R1));
                asm("v_mul_f32 %0, %1, %2" : "=v"(PP) : "v"(P0), "v"(P1));
                asm("v_log_f32 %0, %1" : "=v"(lg) : "v"(PP));
                asm("v_add_f32 %0, %0, %1" : "+v"(L) : "v"(lg));
            }
            L += s0 + s1 + gx + gy;
        } else if constexpr (KIND == K_BODY_PK) {
            // the current FAST body of sweep_pairs2 (dae_triplet.hip), 8 register pairs, one positive
            const float u = b, fp = c;
            f32x2 sgp2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2 t2 = q - u + p[i] * 0.f;      // stands for v2[q] - u  (keeps a per-pair operand)
                const f32x2 w2 = p[i] * fp + 1.0f;
                const float P = w2.x * w2.y;
                const float R = __builtin_amdgcn_rcpf(P);
                L += __builtin_amdgcn_logf(P);
                const f32x2 r2 = f32x2{w2.y, w2.x} * R;
                p[i] += r2 * 1e-9f;
                sgp2 += r2;
                cnt += (unsigned)__popcll(__ballot(t2.x > 1e-16f)) + (unsigned)__popcll(__ballot(t2.y > 1e-16f));
            }
            L += sgp2.x + sgp2.y;
        } else if constexpr (KIND == K_BODY_SC) {
            const float u = b, fp = c;
            float sg = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float tx, ty, wx, wy, P, R, lg, rx, ry;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(tx) : "v"(p[i].x), "v"(u));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ty) : "v"(p[i].y), "v"(u));
                asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(wx) : "v"(p[i].x), "v"(fp));
                asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(wy) : "v"(p[i].y), "v"(fp));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(P) : "v"(wx), "v"(wy));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(R) : "v"(P));
                asm volatile("v_log_f32 %0, %1" : "=v"(lg) : "v"(P));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(L) : "v"(lg));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(rx) : "v"(wy), "v"(R));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ry) : "v"(wx), "v"(R));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(rx), "v"(c));     // gs.x += (stand-in)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i].y) : "v"(ry), "v"(0.f)); // gs.y +=
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sg) : "v"(rx));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sg) : "v"(ry));
                cnt += (unsigned)__popcll(__ballot(tx > 1e-16f)) + (unsigned)__popcll(__ballot(ty > 1e-16f));
            }
            L += sg;
        } else if constexpr (KIND == K_BODY_NEW || KIND == K_BODY_NEW4) {
            // proposed: compare against a per-positive threshold (no t2), log of the product of 2 (or 4) pairs
            const float g = b, fp = c;
            f32x2 sgp2 = {0.f, 0.f};
            constexpr int SH = KIND == K_BODY_NEW ? 2 : 4;
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += SH) {
                float PP = 1.0f;
#pragma unroll
                for (int j = 0; j < SH; ++j) {
                    const int i = i0 + j;
                    const f32x2 w2 = p[i] * fp + 1.0f;
                    const float P = w2.x * w2.y;
                    const float R = __builtin_amdgcn_rcpf(P);
                    PP *= P;
                    const f32x2 r2 = f32x2{w2.y, w2.x} * R;
                    p[i] += r2 * 1e-9f;
                    sgp2 += r2;
                    cnt += (unsigned)__popcll(__ballot(q.x + 0.f * w2.x > g)) + (unsigned)__popcll(__ballot(q.y + 0.f * w2.y > g));
                }
                L += __builtin_amdgcn_logf(PP);
            }
            L += sgp2.x + sgp2.y;
        } else if constexpr (KIND == K_BODY_NEWSC) {
            const float g = b, fp = c;
            float sg = 0.f;
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 2) {
                float PP;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = i0 + j;
                    float wx, wy, P, R, rx, ry;
                    asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(wx) : "v"(p[i].x), "v"(fp));
                    asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(wy) : "v"(p[i].y), "v"(fp));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(P) : "v"(wx), "v"(wy));
                    asm volatile("v_rcp_f32 %0, %1" : "=v"(R) : "v"(P));
                    if (j == 0) PP = P; else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(PP) : "v"(P));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(rx) : "v"(wy), "v"(R));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ry) : "v"(wx), "v"(R));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(rx));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[i].y) : "v"(ry));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(sg) : "v"(rx));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(sg) : "v"(ry));
                    { unsigned n; asm volatile("v_cmp_gt_f32 vcc, %1, %2\n s_bcnt1_i32_b64 %0, vcc" : "=s"(n) : "v"(p[i].x), "v"(g) : "vcc", "scc"); cnt += n; }
                    { unsigned n; asm volatile("v_cmp_gt_f32 vcc, %1, %2\n s_bcnt1_i32_b64 %0, vcc" : "=s"(n) : "v"(p[i].y), "v"(g) : "vcc", "scc"); cnt += n; }
                }
                float lg;
                asm volatile("v_log_f32 %0, %1" : "=v"(lg) : "v"(PP));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(L) : "v"(lg));
            }
            L += sg;
        }
    }
    const long long t1 = clock64();
    float s = L + (float)cnt;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = 1;                       // keeps every chain live
    if (lane == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*probe_fn)(long long*, int, float);
template <int K> static void fill(probe_fn* t) { t[K] = probe<K>; if constexpr (K + 1 < K_COUNT) fill<K + 1>(t); }

int main() {
    probe_fn tab[K_COUNT];
    fill<0>(tab);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    long long* d;
    hipMalloc(&d, (1 + cus * 16) * sizeof(long long));
    std::vector<long long> h(1 + cus * 16);
    const int iters = 2000;
    printf("%-64s %s\n", "instruction / body (cycles per instruction, or per PAIR of triplets)", "waves/SIMD: per-wave | per-SIMD");
    for (int k = 0; k < K_COUNT; ++k) {
        printf("%-64s", kNames[k]);
        for (int W = 1; W <= 4; ++W) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(d, 0, h.size() * sizeof(long long));
                hipLaunchKernelGGL(tab[k], dim3(cus), dim3(256 * W), 0, 0, d, iters, 0.37f);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                std::vector<long long> v;
                for (int b = 0; b < cus; ++b) for (int w = 0; w < 4 * W; ++w) v.push_back(h[1 + b * 16 + w]);
                std::sort(v.begin(), v.end());
                const double med = (double)v[v.size() / 2] / ((double)iters * kPerIter[k]);
                best = std::min(best, med);
            }
            printf("  W%d %6.2f | %5.2f", W, best, best / W);
        }
        printf("\n");
    }
    hipFree(d);
    return 0;
}
