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
    hipLaunchKernelGGL(row_normalize_kernel, dim3(Np), dim3(256), 0, st, X, ldx, N, D, norm, metric == 0 ? 1 : 0, Y, (int64_t)Dp, Dp);
    DAE_CHECK_LAUNCH();
    if (int rc = launch_gemm_f32out(DAE_F32, Np, Np, Y, Dp, Y, Dp, Dp, nullptr, 0, nullptr, 0, 0, out, ldo, 1, 0, st, GEMM_ROLE_GENERIC)) return rc;
    if (zero_diagonal) {
        hipLaunchKernelGGL(zero_diag_kernel, dim3((N + 255) / 256), dim3(256), 0, st, out, ldo, N);
        DAE_CHECK_LAUNCH();
    }
    return 0;
}
