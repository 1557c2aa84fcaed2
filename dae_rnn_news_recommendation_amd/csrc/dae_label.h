// dae_label.h -- label statistics of a mini-batch (triplet_loss_utils.py:47-76,110-111,129), integer exact.
// Shared by dae_elementwise.hip (stand-alone launch) and dae_gather.hip (extra block of the CSR gather).
#pragma once
#include "dae_common.h"
#include "dae_kernels.h"

namespace dae {

constexpr int LABEL_SMEM_BYTES = 4096 + 4096 * 4 + 64;

// single-launch variant for B <= 1024 (the usual mini-batch): ONE block of NT threads, 1024 / NT labels per thread.  Label
// multiplicities come from an LDS histogram when every id lies in [0, 4096) (the Python layer always passes dense
// ids); arbitrary ids fall back to the O(B^2) comparison loop.  Integer arithmetic only -> exact and order-independent.
// Also runs as an extra block of the CSR gather kernel (dae_gather.hip), which saves its own ~5 us launch.
// smem: LABEL_SMEM_BYTES, 16-byte aligned.
template <int NT>
__device__ __forceinline__ void label_stats_block(const LabelJob& j, char* smem) {
    constexpr int NBIN = 4096, E = 1024 / NT;
    int32_t* lab = reinterpret_cast<int32_t*>(smem);                       // [1024]
    int* hist = reinterpret_cast<int*>(smem + 4096);                       // [NBIN]
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem + 4096 + NBIN * 4);   // {S, NV}
    int* out_of_range = reinterpret_cast<int*>(acc + 2);
    const int t = threadIdx.x;
    const int B = j.B, Bp = j.Bp, triplet = j.triplet;
    if (triplet == DAE_TRIPLET_NONE) {
        for (int i = t; i < Bp; i += NT) j.cw[i] = (i < B) ? 1.0f / ((float)B + 1e-16f) : 0.f;
        return;
    }
    if (t == 0) { acc[0] = 0ull; acc[1] = 0ull; *out_of_range = 0; }
    for (int k = t; k < NBIN; k += NT) hist[k] = 0;
    int32_t li[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t + e * NT;
        li[e] = (i < B) ? j.labels[i] : 0;
        lab[i] = (i < B) ? li[e] : (int32_t)0x80000000;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t + e * NT;
        if (i < B) {
            if (li[e] >= 0 && li[e] < NBIN) atomicAdd(&hist[li[e]], 1);
            else *out_of_range = 1;
        }
    }
    __syncthreads();
    long long n[E];
    unsigned s1 = 0u, s2 = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t + e * NT;
        n[e] = 0;
        if (i < B) {
            if (!*out_of_range) {
                n[e] = hist[li[e]];
            } else {
                int cnt = 0;
                const int4* l4 = reinterpret_cast<const int4*>(lab);
                const int n4 = (B + 3) >> 2;
                for (int k = 0; k < n4; ++k) {
                    const int4 v = l4[k];
                    cnt += (v.x == li[e]) + (v.y == li[e]) + (v.z == li[e]) + (v.w == li[e]);
                }
                n[e] = cnt;
            }
            s1 += (unsigned)(n[e] - 1);
            s2 += (unsigned)((n[e] - 1) * (B - n[e]));                     // <= 16 * 64 * 2.7e5 per wave: fits 32 bits
        }
    }
    {   // one LDS atomic per wave, not per thread
        const unsigned w1 = wave_sum_u32(s1), w2 = wave_sum_u32(s2);
        if ((t & 63) == 0) { atomicAdd(&acc[0], (unsigned long long)w1); atomicAdd(&acc[1], (unsigned long long)w2); }
    }
    __syncthreads();
    if (j.order && triplet == DAE_TRIPLET_BATCH_ALL) {
        // Counting sort of the rows by sweep cost (n-1)(B-n), descending.  The cost is a parabola in the class size n, so
        // key = |2n - (B+1)| in [0, B) orders it (smaller key = longer sweep).  Slots inside a bucket are handed out by an LDS
        // atomic: the order among equal-cost anchors is arbitrary, which is fine -- it only decides which workgroup runs first.
        int* cntK = hist;                                                  // [1024]; the label histogram is no longer needed
        for (int k = t; k < 1024; k += NT) cntK[k] = 0;
        __syncthreads();
        int key[E], slot[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t + e * NT;
            key[e] = 0; slot[e] = 0;
            if (i < B) {
                const int d = 2 * (int)n[e] - (B + 1);
                key[e] = d < 0 ? -d : d;
                slot[e] = atomicAdd(&cntK[key[e]], 1);
            }
        }
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                         // inclusive scan of the 1024 bucket counts
            int tmp[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { const int i = t + e * NT; tmp[e] = (i >= off) ? cntK[i - off] : 0; }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) { const int i = t + e * NT; cntK[i] += tmp[e]; }
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t + e * NT;
            if (i < B) j.order[(key[e] > 0 ? cntK[key[e] - 1] : 0) + slot[e]] = i;
        }
    }
    if (j.cls && triplet == DAE_TRIPLET_BATCH_ALL) {
        // class ranges of a label-sorted batch: sortedness by neighbour comparison (block-wide AND through the LDS flag that is
        // free again), the class of row i = [lower_bound, upper_bound) of its label in the sorted LDS copy
        int* flag = out_of_range + 1;
        if (t == 0) *flag = 1;
        __syncthreads();
        bool ok = true;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t + e * NT;
            if (i + 1 < B && lab[i] > lab[i + 1]) ok = false;
        }
        if (!ok) *flag = 0;
        __syncthreads();
        const int sorted = *flag;
        if (t == 0) j.cls[0] = sorted;
        if (sorted) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = t + e * NT;
                if (i < B) {
                    const int32_t v = li[e];
                    int a0 = 0, a1 = i;                 // first index with lab >= v lies in [0, i]
                    while (a0 < a1) { const int m = (a0 + a1) >> 1; if (lab[m] < v) a0 = m + 1; else a1 = m; }
                    j.cls[1 + 2 * i] = a0;
                    j.cls[2 + 2 * i] = a0 + (int)n[e];
                }
            }
        }
    }
    const long long S = (long long)acc[0], NV = (long long)acc[1];
    if (t == 0 && j.nvalid) j.nvalid[0] = NV;
    if (t == 0 && j.tri_scalars && triplet == DAE_TRIPLET_BATCH_ALL) j.tri_scalars[0] = j.alpha / ((float)NV + 1e-16f);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t + e * NT;
        if (i < B) {
            const long long dw = 2 * (n[e] - 1) * (B - n[e]) + (S - n[e] * (n[e] - 1));
            if (j.dw) j.dw[i] = dw;
            if (triplet == DAE_TRIPLET_BATCH_ALL) j.cw[i] = (float)dw / ((float)(3 * NV) + 1e-16f);
        } else if (i < Bp && triplet == DAE_TRIPLET_BATCH_ALL) {
            j.cw[i] = 0.f;
        }
    }
}

}  // namespace dae
