// dae_similarity.hip -- N x N similarity of row vectors on the device: the evaluation step right after the training
// path.  Replaces helpers.pairwise_similarity (helpers.py:11-50), i.e.
//   sklearn.preprocessing.normalize(X, norm) [optional]  ->  cosine_similarity (= l2-normalise rows, X X^T) or
//   linear_kernel (X X^T)  ->  np.fill_diagonal(out, 0),
// called six times per run by main_autoencoder.py:307-317 on the 8000 x 500 embeddings and the 8000 x 10000 BoW / TF-IDF
// matrices.  Rows are normalised once into a zero-padded fp32 image, the product is the exact-fp32 MFMA GEMM of
// dae_gemm.hip (both operands = the same K-contiguous image), the diagonal is cleared in place.
#include "dae_common.h"
#include "dae_kernels.h"

namespace dae {

// one workgroup per row: r = x / n1(x) (norm option), then y = r / ||r||_2 if the metric is cosine.
// sklearn's normalize leaves all-zero rows untouched (norm 0 -> 1).
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ X, int64_t ldx, int N, int D, int norm,
                                                            int cosine, float* __restrict__ Y, int64_t ldy, int Dp) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    float* y = Y + (int64_t)i * ldy;
    if (i >= N) {                                       // padding rows of the operand image
        for (int j = tid; j < Dp; j += 256) y[j] = 0.f;
        return;
    }
    const float* x = X + (int64_t)i * ldx;
    auto block_reduce = [&](float v, bool is_max) {
        v = is_max ? wave_max(v) : wave_sum(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
    };
    float s1 = 1.f;
    if (norm != 0) {
        float a = 0.f;
        for (int j = tid; j < D; j += 256) {
            const float v = x[j];
            a = norm == 1 ? a + fabsf(v) : norm == 2 ? a + v * v : fmaxf(a, fabsf(v));
        }
        a = block_reduce(a, norm == 3);
        if (norm == 2) a = sqrtf(a);
        s1 = a == 0.f ? 1.f : 1.f / a;
    }
    float s2 = 1.f;
    if (cosine) {
        float a = 0.f;
        for (int j = tid; j < D; j += 256) { const float v = x[j] * s1; a += v * v; }
        a = sqrtf(block_reduce(a, false));
        s2 = a == 0.f ? 1.f : 1.f / a;
    }
    for (int j = tid; j < Dp; j += 256) y[j] = j < D ? (x[j] * s1) * s2 : 0.f;
}

__global__ void zero_diag_kernel(float* __restrict__ out, int64_t ldo, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[(int64_t)i * ldo + i] = 0.f;
}

}  // namespace dae

using namespace dae;

extern "C" uint64_t dae_pairwise_similarity_workspace(int32_t N, int32_t D) {
    return (uint64_t)pad128(N) * (uint64_t)pad128(D) * 4ull;
}

extern "C" int dae_pairwise_similarity(const float* X, int64_t ldx, int32_t N, int32_t D, int32_t norm, int32_t metric,
                                       int32_t zero_diagonal, float* out, int64_t ldo, void* workspace, uint64_t workspace_bytes,
                                       void* stream) {
    DAE_CHECK_ARG(X && out && workspace && N > 0 && D > 0 && ldx >= D, "pairwise_similarity: bad input");
    DAE_CHECK_ARG(norm >= 0 && norm <= 3, "pairwise_similarity: norm must be 0 (none), 1 (l1), 2 (l2) or 3 (max)");
    DAE_CHECK_ARG(metric == 0 || metric == 1, "pairwise_similarity: metric must be 0 (cosine) or 1 (linear kernel)");   // helpers.py:34
    const int Np = (int)pad128(N), Dp = (int)pad128(D);
    DAE_CHECK_ARG(ldo >= Np, "pairwise_similarity: out must be a padded image [dae_pad(N) x ldo], ldo >= dae_pad(N) = %d", Np);
    DAE_CHECK_ARG(workspace_bytes >= dae_pairwise_similarity_workspace(N, D), "pairwise_similarity: workspace too small");
    DAE_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "pairwise_similarity: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float* Y = (float*)workspace;
    DAE_LAUNCH(row_normalize_kernel, dim3(Np), dim3(256), 0, st, X, ldx, N, D, norm, metric == 0 ? 1 : 0, Y, (int64_t)Dp, Dp);
    DAE_CHECK_LAUNCH();
    if (int rc = launch_gemm_f32out(DAE_F32, Np, Np, Y, Dp, Y, Dp, Dp, nullptr, 0, nullptr, 0, 0, out, ldo, 1, 0, st, GEMM_ROLE_GENERIC)) return rc;
    if (zero_diagonal) {
        DAE_LAUNCH(zero_diag_kernel, dim3((N + 255) / 256), dim3(256), 0, st, out, ldo, N);
        DAE_CHECK_LAUNCH();
    }
    return 0;
}

// =================================================================================================================
// Related / unrelated pair statistics of a similarity matrix.
// Replaces the numeric part of helpers.visualize_pairwise_similarity (helpers.py:79-135): pairs (i, j), j < i, whose two labels
// are both >= 0 are "related" when the labels are equal and "unrelated" otherwise; the reference hands the two score lists
// to sklearn's roc_curve / auc (on Python lists of 3.2e7 elements) and draws a box plot of them.  Here:
//   1. one pass over the strict lower triangle splits the scores into two key arrays (order-preserving uint32 keys);
//   2. rocPRIM device radix sort of each array (library sort, like hipBLASLt for a plain GEMM);
//   3. AUROC = P(related > unrelated) + 0.5 P(equal): every related key is located in the sorted unrelated keys by two
//      binary searches, (lower + upper) summed in 64-bit integers -- exact, tie-aware, order-independent;
//   4. the box-plot numbers (min, quartiles, max, mean) are order statistics of the sorted arrays.
// =================================================================================================================
#include <rocprim/rocprim.hpp>

#include <cmath>
#include <vector>

namespace dae {

__device__ __forceinline__ uint32_t score_key(float f) {      // monotone: a < b  <=>  key(a) < key(b); -0 < +0 adjacent
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// one workgroup per row i: columns j < i.  Appends with one atomic per wave and class (ballot-aggregated).
__global__ __launch_bounds__(256) void split_pairs_kernel(const float* __restrict__ S, int64_t lds, const int32_t* __restrict__ labels,
                                                         int N, uint32_t* __restrict__ rel, uint32_t* __restrict__ unrel,
                                                         unsigned long long* __restrict__ counters) {
    const int i = blockIdx.x + 1;
    if (i >= N) return;
    const int32_t li = labels[i];
    if (li < 0) return;
    const int lane = threadIdx.x & 63;
    for (int j0 = 0; j0 < i; j0 += 256) {
        const int j = j0 + threadIdx.x;
        int32_t lj = -1;
        float s = 0.f;
        if (j < i) { lj = labels[j]; s = S[(int64_t)i * lds + j]; }
        const bool valid = lj >= 0;
        const bool is_rel = valid && lj == li, is_un = valid && lj != li;
        const unsigned long long br = __ballot(is_rel), bu = __ballot(is_un);
        unsigned long long base_r = 0, base_u = 0;
        if (lane == 0) {
            if (br) base_r = atomicAdd(&counters[0], (unsigned long long)__popcll(br));
            if (bu) base_u = atomicAdd(&counters[1], (unsigned long long)__popcll(bu));
        }
        base_r = __shfl(base_r, 0, 64); base_u = __shfl(base_u, 0, 64);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (is_rel) rel[base_r + __popcll(br & lt)] = score_key(s);
        if (is_un) unrel[base_u + __popcll(bu & lt)] = score_key(s);
    }
}

// for every related key: #unrelated strictly below + #unrelated not above  (= 2 * "wins" with ties counted half)
__global__ __launch_bounds__(256) void auroc_count_kernel(const uint32_t* __restrict__ rel, unsigned long long n_rel,
                                                         const uint32_t* __restrict__ unrel, unsigned long long n_un,
                                                         unsigned long long* __restrict__ part) {
    __shared__ unsigned long long red[4];
    unsigned long long acc = 0;
    for (unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x; t < n_rel; t += (unsigned long long)gridDim.x * 256) {
        const uint32_t k = rel[t];
        unsigned long long lo = 0, hi = n_un;                 // first index with unrel >= k
        while (lo < hi) { const unsigned long long m = (lo + hi) >> 1; if (unrel[m] < k) lo = m + 1; else hi = m; }
        const unsigned long long lower = lo;
        hi = n_un;                                            // first index with unrel > k
        while (lo < hi) { const unsigned long long m = (lo + hi) >> 1; if (unrel[m] <= k) lo = m + 1; else hi = m; }
        acc += lower + lo;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// fixed-shape sum of the scores of a sorted key array: one double per block, added up on the host in block order
__global__ __launch_bounds__(256) void key_sum_kernel(const uint32_t* __restrict__ keys, unsigned long long n, double* __restrict__ part) {
    __shared__ double red[256];
    double acc = 0.0;
    for (unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * 256)
        acc += (double)key_score(keys[t]);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void pick_keys_kernel(const uint32_t* __restrict__ keys, const unsigned long long* __restrict__ idx, int n, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = key_score(keys[idx[t]]);
}

static size_t sort_temp_bytes(size_t n) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, bytes, p, p, n, 0, 32, (hipStream_t)0);
    return bytes;
}
static inline uint64_t al256(uint64_t b) { return (b + 255) / 256 * 256; }

}  // namespace dae

constexpr int PAIR_PART_BLOCKS = 1024;

extern "C" uint64_t dae_pair_stats_workspace(int32_t N) {
    const uint64_t npairs = (uint64_t)N * (uint64_t)(N > 0 ? N - 1 : 0) / 2;
    // worst case per class = all pairs: unsorted + sorted keys of both classes, sort scratch, labels, counters, partials, picks
    return 4 * al256(npairs * 4) + al256(sort_temp_bytes((size_t)npairs)) + al256((uint64_t)N * 4) + 256 +
           al256(PAIR_PART_BLOCKS * 8) * 3 + 4096;
}

extern "C" int dae_pair_stats(const float* S, int64_t lds, const int32_t* labels_host, int32_t N, double* out16, void* workspace,
                              uint64_t workspace_bytes, void* stream) {
    DAE_CHECK_ARG(S && labels_host && out16 && workspace && N > 1 && lds >= N, "pair_stats: bad arguments");
    DAE_CHECK_ARG(workspace_bytes >= dae_pair_stats_workspace(N) && ((uintptr_t)workspace % 256) == 0, "pair_stats: workspace too small / unaligned");
    hipStream_t st = (hipStream_t)stream;
    // class sizes from the label histogram (helpers.py:91-96: labels < 0 are missing values)
    uint64_t n_valid = 0, n_rel = 0;
    {
        std::vector<int32_t> l(labels_host, labels_host + N);
        std::vector<int32_t> s;
        for (int32_t v : l) if (v >= 0) s.push_back(v);
        n_valid = s.size();
        std::sort(s.begin(), s.end());
        for (size_t a = 0; a < s.size();) {
            size_t b = a;
            while (b < s.size() && s[b] == s[a]) ++b;
            n_rel += (uint64_t)(b - a) * (uint64_t)(b - a - 1) / 2;
            a = b;
        }
    }
    const uint64_t n_un = n_valid * (n_valid > 0 ? n_valid - 1 : 0) / 2 - n_rel;
    const uint64_t npairs = (uint64_t)N * (uint64_t)(N - 1) / 2;
    char* w = (char*)workspace;
    uint32_t* rel = (uint32_t*)w;            w += al256(npairs * 4);
    uint32_t* unrel = (uint32_t*)w;          w += al256(npairs * 4);
    uint32_t* rel_s = (uint32_t*)w;          w += al256(npairs * 4);
    uint32_t* unrel_s = (uint32_t*)w;        w += al256(npairs * 4);
    const size_t temp_bytes = sort_temp_bytes((size_t)npairs);
    void* temp = w;                          w += al256(temp_bytes);
    int32_t* labels_dev = (int32_t*)w;       w += al256((uint64_t)N * 4);
    unsigned long long* counters = (unsigned long long*)w;   w += 256;
    unsigned long long* part_u = (unsigned long long*)w;     w += al256(PAIR_PART_BLOCKS * 8);
    double* part_r = (double*)w;             w += al256(PAIR_PART_BLOCKS * 8);
    double* part_n = (double*)w;             w += al256(PAIR_PART_BLOCKS * 8);
    unsigned long long* pick_idx = (unsigned long long*)w;   w += 2048;
    float* pick_val = (float*)w;
    for (int k = 0; k < 16; ++k) out16[k] = std::nan("");
    out16[1] = (double)n_rel; out16[2] = (double)n_un;
    DAE_CHECK_HIP(hipMemcpyAsync(labels_dev, labels_host, (size_t)N * 4, hipMemcpyHostToDevice, st));
    DAE_CHECK_HIP(hipMemsetAsync(counters, 0, 16, st));
    DAE_LAUNCH(split_pairs_kernel, dim3(N - 1), dim3(256), 0, st, S, lds, labels_dev, N, rel, unrel, counters);
    DAE_CHECK_LAUNCH();
    size_t tb = temp_bytes;
    if (n_rel) DAE_CHECK_HIP(rocprim::radix_sort_keys(temp, tb, rel, rel_s, (size_t)n_rel, 0, 32, st));
    tb = temp_bytes;
    if (n_un) DAE_CHECK_HIP(rocprim::radix_sort_keys(temp, tb, unrel, unrel_s, (size_t)n_un, 0, 32, st));
    unsigned long long cnt[2] = {0, 0};
    DAE_CHECK_HIP(hipMemcpyAsync(cnt, counters, 16, hipMemcpyDeviceToHost, st));
    if (n_rel && n_un) {
        DAE_LAUNCH(auroc_count_kernel, dim3(PAIR_PART_BLOCKS), dim3(256), 0, st, rel_s, (unsigned long long)n_rel, unrel_s,
                           (unsigned long long)n_un, part_u);
        DAE_CHECK_LAUNCH();
    }
    // order statistics: numpy's default (linear) percentile -> positions q * (n - 1)
    unsigned long long idx_h[32];
    int npick = 0;
    auto add_class = [&](uint64_t n, uint64_t base) {
        const double q[5] = {0.0, 0.25, 0.5, 0.75, 1.0};
        for (int k = 0; k < 5; ++k) {
            const double pos = q[k] * (double)(n - 1);
            const uint64_t lo = (uint64_t)std::floor(pos), hi = (uint64_t)std::ceil(pos);
            idx_h[npick++] = base + lo; idx_h[npick++] = base + hi;
        }
    };
    // the two sorted arrays are addressed through one index space: [rel_s | unrel_s] are not contiguous, so pick separately
    std::vector<float> pv(20, 0.f);
    if (n_rel) {
        npick = 0; add_class(n_rel, 0);
        DAE_CHECK_HIP(hipMemcpyAsync(pick_idx, idx_h, 10 * 8, hipMemcpyHostToDevice, st));
        DAE_LAUNCH(pick_keys_kernel, dim3(1), dim3(32), 0, st, rel_s, pick_idx, 10, pick_val);
        DAE_CHECK_LAUNCH();
        DAE_LAUNCH(key_sum_kernel, dim3(PAIR_PART_BLOCKS), dim3(256), 0, st, rel_s, (unsigned long long)n_rel, part_r);
        DAE_CHECK_LAUNCH();
        DAE_CHECK_HIP(hipMemcpyAsync(pv.data(), pick_val, 10 * 4, hipMemcpyDeviceToHost, st));
        DAE_CHECK_HIP(hipStreamSynchronize(st));              // idx_h / pick buffers are reused for the second class
    }
    if (n_un) {
        npick = 0; add_class(n_un, 0);
        DAE_CHECK_HIP(hipMemcpyAsync(pick_idx, idx_h, 10 * 8, hipMemcpyHostToDevice, st));
        DAE_LAUNCH(pick_keys_kernel, dim3(1), dim3(32), 0, st, unrel_s, pick_idx, 10, pick_val);
        DAE_CHECK_LAUNCH();
        DAE_LAUNCH(key_sum_kernel, dim3(PAIR_PART_BLOCKS), dim3(256), 0, st, unrel_s, (unsigned long long)n_un, part_n);
        DAE_CHECK_LAUNCH();
        DAE_CHECK_HIP(hipMemcpyAsync(pv.data() + 10, pick_val, 10 * 4, hipMemcpyDeviceToHost, st));
    }
    std::vector<unsigned long long> pu(PAIR_PART_BLOCKS, 0);
    std::vector<double> pr(PAIR_PART_BLOCKS, 0.0), pn(PAIR_PART_BLOCKS, 0.0);
    if (n_rel && n_un) DAE_CHECK_HIP(hipMemcpyAsync(pu.data(), part_u, PAIR_PART_BLOCKS * 8, hipMemcpyDeviceToHost, st));
    if (n_rel) DAE_CHECK_HIP(hipMemcpyAsync(pr.data(), part_r, PAIR_PART_BLOCKS * 8, hipMemcpyDeviceToHost, st));
    if (n_un) DAE_CHECK_HIP(hipMemcpyAsync(pn.data(), part_n, PAIR_PART_BLOCKS * 8, hipMemcpyDeviceToHost, st));
    DAE_CHECK_HIP(hipStreamSynchronize(st));
    DAE_CHECK_ARG(cnt[0] == n_rel && cnt[1] == n_un, "pair_stats: class sizes from the labels (%llu, %llu) and from the split pass (%llu, %llu) differ",
                  (unsigned long long)n_rel, (unsigned long long)n_un, cnt[0], cnt[1]);
    if (n_rel && n_un) {
        long double twice = 0.0L;
        for (unsigned long long v : pu) twice += (long double)v;
        out16[0] = (double)(twice / (2.0L * (long double)n_rel * (long double)n_un));
    }
    auto fill = [&](uint64_t n, const std::vector<double>& parts, const float* picks, int o_mean, int o_five) {
        if (!n) return;
        double sum = 0.0;
        for (double v : parts) sum += v;
        out16[o_mean] = sum / (double)n;
        const double q[5] = {0.0, 0.25, 0.5, 0.75, 1.0};
        for (int k = 0; k < 5; ++k) {
            const double pos = q[k] * (double)(n - 1), fr = pos - std::floor(pos);
            out16[o_five + k] = (double)picks[2 * k] + ((double)picks[2 * k + 1] - (double)picks[2 * k]) * fr;
        }
    };
    fill(n_rel, pr, pv.data(), 3, 5);
    fill(n_un, pn, pv.data() + 10, 4, 10);
    return 0;
}
