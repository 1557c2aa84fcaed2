/*
 * dae_hip.h -- C ABI of libdae_hip.so: the MI355X (gfx950) implementation of the DAE
 * article-embedding training hot path of louislung/DAE_RNN_News_Recommendation.
 *
 * The reference has NO native/FFI boundary (100 % Python on TensorFlow 1.12): the hot path is
 * whatever `tf_session.run([train_step, ...])` executes per mini-batch.  This header therefore
 * defines the boundary a maintainer would bind with ctypes from the reference's own Python
 * (see INTEGRATION.md).  Each entry point cites the reference code it replaces; paths are
 * relative to the reference repo root.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - every function returns 0 on success, non-zero on error (message via dae_last_error());
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     stream-ordered, nothing synchronises the host;
 *   - dense activations use one element type per call: DAE_BF16 (MFMA bf16, fp32 accumulate)
 *     or DAE_F32 (exact-fp32 MFMA, the parity mode); parameters / gradients / statistics are fp32;
 *   - "padded" sizes: every dense row/column extent is rounded up to a multiple of DAE_PAD (128)
 *     and the padding is kept EXACTLY ZERO by every kernel (see DESIGN.md "HBM layout").
 */
#ifndef DAE_HIP_H
#define DAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAE_PAD 128
#define DAE_ABI_VERSION 6   /* 6: dae_comm_* / dae_allreduce_grads / dae_dp_exchange / dae_dp_bands (the data-parallel collective in the C ABI, RCCL on the step's stream); 5: dae_storage_format (the fp16 build libdae_hip_f16.so), options x3_terms / op_scale_log2, DAE_WAIT_DW_CREATED = 100; 4: DAE_BF16X3 (split-bf16 mode), dae_gemm_nt_n; 3: dae_buffers.grad_lo, options dw_bits / encode_w32; 2: dae_step.c_row_idx, plan options, phases 4/5, sharded apply */

enum { DAE_BF16 = 0, DAE_F32 = 1,
       DAE_BF16X3 = 2 /* dae_config.dtype only: bf16 storage and MFMA, but every stored operand of the three gradient GEMMs is kept as
                         hi + lo (both bf16) and multiplied as (hi,hi) + (hi,lo) + (lo,hi) -- 2^-16 operands at ~3x the bf16 GEMM work;
                         every input kind and phase.  In the fp16 build of the library (dae_storage_format() == 1) the same dtype keeps only the
                         lo terms of plan option "x3_terms" -- by default the two W terms -- i.e. TWO products per gradient GEMM: the product's
                         precision='f16x2' */ };
enum { DAE_ACT_NONE = 0, DAE_ACT_SIGMOID = 1, DAE_ACT_TANH = 2 };
enum { DAE_LOSS_CROSS_ENTROPY = 0, DAE_LOSS_MEAN_SQUARED = 1, DAE_LOSS_COSINE = 2 };
enum { DAE_OPT_SGD = 0, DAE_OPT_ADAGRAD = 1, DAE_OPT_MOMENTUM = 2, DAE_OPT_ADAM = 3 };
enum { DAE_TRIPLET_NONE = 0, DAE_TRIPLET_BATCH_ALL = 1, DAE_TRIPLET_BATCH_HARD = 2,
       DAE_TRIPLET_EXPLICIT = 3 /* DenoisingAutoencoderTriplet: rows stacked [org; pos; neg] */ };
enum { DAE_CORR_NONE = 0, DAE_CORR_KEEPBITS = 1, DAE_CORR_PHILOX_MASK = 2 };
/* `mode` bits of dae_triplet_batch_all */
enum { DAE_MINER_POS_ONLY = 1, DAE_MINER_FAST = 2 };

/* slots of the per-step statistics record (float[DAE_STATS_STRIDE]) -- the values the reference
 * fetches at autoencoder.py:233 and averages per epoch at :283-294 */
enum { DAE_STAT_COST = 0, DAE_STAT_AE = 1, DAE_STAT_TRIPLET = 2, DAE_STAT_FRACTION = 3,
       DAE_STAT_NUM = 4, DAE_STAT_NVALID = 5, DAE_STATS_STRIDE = 8 };

int         dae_abi_version(void);
const char* dae_last_error(void);
/* padded extent: ceil(n / DAE_PAD) * DAE_PAD */
int64_t     dae_pad(int64_t n);

/* ---------------------------------------------------------------------------------------------
 * K0+K1 (front half): corrupt + CSR-row -> dense tile gather, staged through LDS.
 * Replaces utils.masking_noise (utils.py:94-115), the per-batch CSR fancy-index
 * (utils.py:59-60), get_sparse_ind_val_shape (utils.py:162-180) and tf.sparse.to_dense
 * (triplet_loss_utils.py:264).
 *   rows  i = 0..B-1 take CSR row row_idx[i];  columns are the CSR's (sorted) column ids.
 *   x      [Bp x ldx]  clean rows (target of the reconstruction loss)          (may be NULL)
 *   xc     [Bp x ldx]  corrupted rows  x~ = scale * keep(e) * value(e)         (may be NULL)
 *   xct    [Fp x ldt]  transpose of xc (A operand of the dW GEMM)              (may be NULL;
 *                      must be zero-filled by the caller, only kept entries are written)
 *   rowsq  [Bp] fp32   sum of squares of each clean row (cosine_proximity)     (may be NULL)
 *   corr_mode: DAE_CORR_NONE      -> keep everything
 *              DAE_CORR_KEEPBITS  -> keep entry e iff bit e of keep_bits (reference-exact stream
 *                                    np.random.rand(nnz) >= v, generated on the host)
 *              DAE_CORR_PHILOX_MASK -> keep iff philox_uniform(e; seed, stream) >= corr_frac
 *   values == NULL means a binary matrix (all stored values 1.0; main_autoencoder.py:235).
 * ------------------------------------------------------------------------------------------- */
int dae_gather_csr(const int64_t* indptr, const int32_t* indices, const float* values,
                   const int32_t* row_idx, int32_t B, int32_t F, int32_t dtype,
                   void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq,
                   int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed, uint32_t rng_stream,
                   float corr_frac, float scale, void* stream);

/* Same kernel, additionally emitting the corrupted batch as a BIT image for dae_encode_bits:
 *   xc_bits [Bp x ldw] u32, bit b of word w of row i  <=>  entry (i, 32*w + b) of x~ is kept (value 1.0).
 * Only for binary matrices (values == NULL) with scale == 1; rows >= B and columns >= F are zero.
 * xc may then be NULL (the dense x~ image is not needed by the encode GEMM). */
int dae_gather_csr_bits(const int64_t* indptr, const int32_t* indices, const float* values,
                        const int32_t* row_idx, int32_t B, int32_t F, int32_t dtype,
                        void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq,
                        int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed, uint32_t rng_stream,
                        float corr_frac, float scale, uint32_t* xc_bits, int64_t ldw, void* stream);

/* K1 for binary inputs: fused corrupt+encode GEMM  slab[s] = bits(x~) . Wt_lo^T  over K range s
 * (autoencoder.py:389 tf.sparse.matmul(sparse x~, W); the reference multiplies the 0/1 CSR directly).
 * The A operand is the bit image and is expanded to bf16 MFMA fragments inside LDS; only W^T is streamed.
 * bf16 only.  Bp, Hp multiples of 128; Fp multiple of 64; slabs as for dae_gemm_nt. */
int dae_encode_bits(const uint32_t* xc_bits, int64_t ldw, const void* Wt_lo, int64_t ldwt,
                    int32_t Bp, int32_t Hp, int32_t Fp, float* slabs, int64_t ld_slab,
                    int32_t splits, int64_t slab_stride, void* stream);

/* K0+K1+K2 for CSR inputs in ONE launch: corrupt + gather + encode on the stored entries -- the reference's own
 * formulation, tf.sparse.matmul(x~, W) + b_h -> activation -> - act(b_h) (autoencoder.py:377,389):
 *     h[i,:] = act( sum_{e in row_idx[i], kept(e)} scale * v_e * W_lo[col_e, :] + bh ) - act(bh)
 * The dense x~ image is never formed (a row holds ~2 % of the features): the kernel reads ~140 W rows per batch row and is
 * bound by L2 -> register bandwidth, not by MFMA.  W_lo: [Fp x ldw] row-major in `dtype` (the bf16 shadow, or fp32 in parity
 * mode); fp32 accumulation in stored-entry order.  Outputs (any may be NULL): h_f32 / h_lo [Bp x ldh], h_t [Hp x ldht],
 * hcat_a / hcat_b split-bf16 Gram operands [Bp x 3Hp]; side images of the batch: x_bits [Bp x ldxb] (bit image of the CLEAN
 * rows, binary data only), xct [Fp x ldt] (kept entries scattered into the pre-zeroed x~^T), rowsq [Bp] (sum of squares of
 * the clean row).  Corruption arguments as for dae_gather_csr. */
int dae_encode_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx,
                   int32_t B, int32_t F, int32_t H, int32_t dtype, const void* W_lo, int64_t ldw, const float* bh,
                   int32_t enc_act, int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed, uint32_t rng_stream,
                   float corr_frac, float scale, float* h_f32, void* h_lo, int64_t ldh, void* h_t, int64_t ldht,
                   void* hcat_a, void* hcat_b, uint32_t* x_bits, int64_t ldxb, void* xct, int64_t ldt, float* rowsq,
                   void* stream);

/* Salt-and-pepper corruption of a mini-batch ON THE DEVICE (utils.salt_and_pepper_noise, utils.py:118-144, with a counter
 * RNG instead of the host stream): for batch row i (train-set row row_idx[i]) `v` column ids are drawn with replacement,
 * col_t = floor(u * F), and each is set to `lo` (coin < 0.5) or `hi`; later draws win.  (u, coin) = the first two words of
 * Philox4x32-10 at counter (row_idx[i], t, rng_stream, 1), key = seed -- restated by oracle.salt_and_pepper_philox.
 * Output: a batch-local CSR with a fixed row capacity `cap` (>= longest row + v): row i occupies out_indices / out_values
 * [i*cap, i*cap + len_i), sorted by column, zeros dropped; out_span[2i], out_span[2i+1] = its start and end, so the arrays
 * serve as (c_indptr = out_span, c_row_idx[i] = 2i) of dae_train_step. */
int dae_salt_pepper_batch(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx,
                          int32_t B, int32_t F, int32_t v, float lo, float hi, uint64_t seed, uint32_t rng_stream,
                          int64_t* out_span, int32_t* out_indices, float* out_values, int32_t cap, void* stream);

/* Reference-exact masking noise on the HOST, continuing NumPy's legacy global stream natively (autoencoder/utils.py:111
 * `np.random.rand(nnz) >= v`; dense input utils.py:108 `np.random.choice([0,1], size, p=[v,1-v])` consumes the same doubles):
 * key[624] / *pos are the ('MT19937', key, pos, ...) fields of np.random.get_state(); n doubles are drawn (2 words each, a>>5
 * and b>>6 as randomkit does) and bit e of bits_out (little-endian uint32 words, (n+31)/32 of them) = (double_e >= corr_frac),
 * decided by an exact integer comparison.  key / *pos come back advanced: np.random.set_state with them continues the stream as
 * if NumPy had drawn the n doubles.  HOST pointers; no stream; thread-safe for distinct states.  Returns 0, or 1 on bad input. */
int dae_host_mt19937_keep_bits(uint32_t* key, int32_t* pos, int64_t n, double corr_frac, uint32_t* bits_out);

/* Dense-ndarray input (autoencoder.py:143 sparse_input=False; utils.py:107-109 dense masking):
 * gathers fp32 rows data[row_idx[i], :] into x / xc / xct (ldx, ldt multiples of 128; x / xc / xct 16-byte aligned) with the
 * keep decision from keep_bits (DAE_CORR_KEEPBITS) or from Philox (DAE_CORR_PHILOX_MASK): element (row, f) is kept iff word
 * f & 3 of Philox4x32-10 at counter (f >> 2, row, rng_stream, 2), key = seed, scaled to [0, 1) by (w >> 8) * 2^-24, is
 * >= corr_frac -- one draw per four neighbouring features (restated by oracle.philox_uniform_dense). */
int dae_gather_dense(const float* data, int64_t ld_data, const int32_t* row_idx, int32_t B, int32_t F,
                     int32_t dtype, void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq,
                     float* rowsq_scratch /* [(Fp/64) x Bp], needed iff rowsq */,
                     int32_t corr_mode, const uint32_t* keep_bits /* bit index = row*F + f */,
                     uint64_t seed, uint32_t rng_stream, float corr_frac, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Generic NT GEMM on MFMA tiles:  C[M x N] (+)= sum_seg A_seg[M x K_seg] * Bt_seg[N x K_seg]^T
 * fp32 accumulate, fp32 output; up to two K segments (tied-weight gradient, triplet term).
 * M, N multiples of 128, K_seg multiples of 64 (bf16) / 32 (fp32); lda/ldb/ldc in ELEMENTS.
 * splits > 1 writes `splits` partial slabs C + s*slab_stride (consumer sums them).
 * This is tf.matmul / tf.sparse.matmul of the reference graph (autoencoder.py:389,411;
 * triplet_loss_utils.py:93,219) and of its autodiff (autoencoder.py:452-472).
 * ------------------------------------------------------------------------------------------- */
int dae_gemm_nt(int32_t dtype, int32_t M, int32_t N,
                const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0, int32_t K0,
                const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int32_t K1,
                float* C, int64_t ldc, int32_t splits, int64_t slab_stride, void* stream);

/* The same contraction over 1..5 K segments: C = sum_s A_s[M x K_s] . Bt_s[N x K_s]^T.  Split-bf16 operands (x = hi + lo, both bf16)
 * multiply as (hi,hi) + (hi,lo) + (lo,hi): three segments per product, the lo.lo term (2^-16 of a 2^-8 term) is dropped.  Segments with
 * K = 0 are skipped.  Same tiles, split-K slabs and kernels as dae_gemm_nt. */
typedef struct dae_gemm_seg { const void* A; int64_t lda; const void* Bt; int64_t ldb; int32_t K; } dae_gemm_seg;
int dae_gemm_nt_n(int32_t dtype, int32_t M, int32_t N, const dae_gemm_seg* segs, int32_t nsegs, float* C, int64_t ldc,
                  int32_t splits, int64_t slab_stride, void* stream);

/* Diagnostic twin of dae_gemm_nt (bf16, LDS-DMA ring depth nst = 2 or 3): same arithmetic, and every wave also writes
 * shader-clock sums of its K-loop phases to trace[(block*4 + wave)*8 + k]:
 *   k=0 first MFMA half (+DMA issue)  1 waits (vmcnt/lgkmcnt)  2 barrier  3 reads + second MFMA half + DMA issue  4 K iterations
 *   5 whole K loop  6 epilogue stores  7 s_memtime at entry.    Used by tools/gemm_trace.py; not on the training path. */
int dae_gemm_trace(int32_t dtype, int32_t M, int32_t N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0,
                   int32_t K0, const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int32_t K1, float* C,
                   int64_t ldc, int32_t splits, int64_t slab_stride, int32_t nst, uint64_t* trace, void* stream);

/* K2: encode epilogue  h = act(sum_s slab_s + bh) - act(bh)   (autoencoder.py:389)
 * writes h fp32 [Bp x ldh], h in `dtype` [Bp x ldh] and h^T in `dtype` [Hp x ldht]; rows >= B
 * and columns >= H are written as zero.  Any output pointer may be NULL.
 * hcat_a / hcat_b (both or neither): bf16 [Bp x 3*Hp] = [hi|hi|lo] and [hi|lo|hi] with h = hi + lo, the
 * operands of the split-bf16 Gram matrix  D ~= hcat_a . hcat_b^T  (one dae_gemm_nt call with K = 3*Hp). */
int dae_encode_finish(const float* slabs, int32_t splits, int64_t slab_stride, int64_t ld_slab,
                      const float* bh, int32_t B, int32_t H, int32_t enc_act, int32_t dtype,
                      float* h_f32, void* h_lo, int64_t ldh, void* h_t, int64_t ldht,
                      void* hcat_a, void* hcat_b, void* stream);

/* K3+K4 (+ seeds of K8): decode GEMM fused with bias, activation, per-row reconstruction loss
 * and d cost / d z2   (autoencoder.py:411; triplet_loss_utils.py:262-277 weighted_loss).
 *   y = act(h W^T + bv);  rowloss_i = sum_f loss(x_if, y_if);
 *   delta2_if = cw_i * dloss/dy * act'(z2)      with cw_i = w_i / (sum w + 1e-16)
 * outputs: rowloss_part [n_col_waves x Bp] partial row sums (n_col_waves = 2*Fp/dae_decode_tile_n(dtype); may be NULL),
 *          tile_part [(Bp/128)*(Fp/dae_decode_tile_n(dtype))] each tile's share of sum_i cw_i * rowloss_i (may be NULL),
 *          dbv_part [n_row_waves x Fp] partial column sums of delta2 (n_row_waves = 2*Bp/128),
 *          delta2 [Bp x ldd] and delta2^T [Fp x lddt] in `dtype` (any of these may be NULL).
 * cosine_proximity needs whole-row statistics and runs as two passes over the same GEMM:
 *   cos_pass 1: cos_stats[0..Bp) = sum x^2 (from the gather) is read, cos_part
 *               [2 x n_col_waves x Bp] receives partial {sum y^2, sum xhat.y};
 *   dae_cos_reduce folds them into cos_stats[Bp..3Bp) and the row loss;
 *   cos_pass 2: produces delta2 / delta2^T / dbv_part.   Other losses: cos_pass = 0. */
int dae_decode_loss(int32_t dtype, int32_t B, int32_t F, int32_t H,
                    const void* h_lo, int64_t ldh, const void* W_lo, int64_t ldw,
                    const float* bv, const void* x, int64_t ldx, const float* cw,
                    int32_t dec_act, int32_t loss_func, int32_t cos_pass, const float* cos_stats,
                    float* cos_part, float* rowloss_part, float* tile_part, float* dbv_part,
                    void* delta2, int64_t ldd, void* delta2_t, int64_t lddt, void* stream);
/* columns of y per workgroup of dae_decode_loss for `dtype` (64 for DAE_BF16, 128 for DAE_F32): rowloss_part / cos_part hold
 * 2 * Fp / width partial rows, tile_part (Bp/128) * (Fp/width) entries */
int32_t dae_decode_tile_n(int32_t dtype);

int dae_cos_reduce(const float* cos_part, int32_t n_col_waves, int32_t B, int32_t Bp,
                   float* cos_stats, float* rowloss, void* stream);

/* K5: Gram matrix D = h h^T in exact fp32 MFMA (triplet_loss_utils.py:93,219).
 * D_slabs: `splits` slabs of [Bp x Bp] (consumers sum them). */
int dae_gram(const float* h_f32, int64_t ldh, int32_t Bp, int32_t Hp, float* D_slabs, int32_t splits,
             void* stream);

/* Label statistics (triplet_loss_utils.py:47-76,110-111,129): integer-exact N_valid and batch_all
 * data_weight from label multiplicities, and cw_i = w_i/(sum w + 1e-16) (zero beyond B).
 *   labels int32[B]; n_same_scratch int32[B]; acc_scratch uint64[2]; nvalid_out int64[1];
 *   dw_out int64[B] (may be NULL); cw float[Bp].
 * DAE_TRIPLET_NONE writes cw_i = 1/(B + 1e-16) (weighted_loss's default weight ones, :266) and
 * needs no labels/scratch; DAE_TRIPLET_BATCH_HARD only fills nvalid/dw (cw comes from the miner). */
int dae_label_stats(const int32_t* labels, int32_t B, int32_t Bp, int32_t triplet,
                    int32_t* n_same_scratch, uint64_t* acc_scratch,
                    int64_t* nvalid_out, int64_t* dw_out, float* cw,
                    float alpha, float* tri_scalars /* may be NULL; batch_all: [0] = alpha/(N_valid+1e-16) */,
                    void* stream);

/* K6: batch_all online miner, one fused sweep per anchor (triplet_loss_utils.py:79-131).
 *   loss_part[a] = sum_{p,n valid} softplus(D[a,n]-D[a,p])   (only positive triplets if pos_only)
 *   npos_part[a] = #{valid (p,n): D[a,n]-D[a,p] > 1e-16}
 *   G[a,:]       = d(sum softplus)/dD[a,:]   (un-normalised, [Bp x Bp], row stride Bp)
 *   role_cnt[a,:] (pos_only): per-column positive-triplet counts (NULL otherwise)
 * D is given as `d_splits` slabs (stride slab_stride) that are summed on load.
 * mode: DAE_MINER_POS_ONLY (the reference's pos_triplets_only=True) | DAE_MINER_FAST (bf16 training mode: sigmoid as
 *       1 - 1/(1+e) and log(fl(1+e)) without the log1p correction -- per-anchor error bound 2e-6 relative, enforced by an
 *       in-kernel fallback to the exact form; counts stay bit-exact). */
int dae_triplet_batch_all(const float* D_slabs, int32_t d_splits, int64_t slab_stride, int64_t ldd,
                          const int32_t* labels, int32_t B, int32_t Bp, int32_t mode,
                          float* loss_part, uint32_t* npos_part, float* G, uint32_t* role_cnt,
                          void* stream);

/* The same miner for the anchors [a0, a0 + n_anchors) of the batch only: row r of D_slabs / G / loss_part / npos_part belongs
 * to batch element a0 + r, D is [n_anchors x ldd] (its columns are the WHOLE batch of B rows, labels[B] likewise).  This is what
 * a data-parallel rank runs on ITS rows against the all-gathered embeddings (global-batch mining, SURVEY 8e mode i). */
int dae_triplet_batch_all_rows(const float* D_slabs, int32_t d_splits, int64_t slab_stride, int64_t ldd,
                               const int32_t* labels, int32_t B, int32_t Bp, int32_t a0, int32_t n_anchors, int32_t mode,
                               float* loss_part, uint32_t* npos_part, float* G, uint32_t* role_cnt, void* stream);

/* K7: batch_hard online miner (triplet_loss_utils.py:202-259) incl. its quirks.
 *   dist_a = max(hn_a - hp_a, 0); cnt_a = dist_a > 0;
 *   loss_part[a] = softplus(dist_a)*cnt_a;  cnt_part[a] = cnt_a;
 *   dw[j] += cnt_a*([D[a,j]==hp_a] + [D[a,j]==hn_a] + [j==a])     (int32 atomics; zeroed here)
 *   G[a,:] = d(sum_a softplus(dist_a) cnt_a)/dD[a,:]  (un-normalised; ties split equally) */
int dae_triplet_batch_hard(const float* D_slabs, int32_t d_splits, int64_t slab_stride, int64_t ldd,
                           const int32_t* labels, int32_t B, int32_t Bp,
                           float* loss_part, uint32_t* cnt_part, int32_t* dw, float* G, void* stream);

/* batch_hard for the anchors [a0, a0 + n_anchors) only (see dae_triplet_batch_all_rows).  dw[B] is zeroed by EVERY call and then
 * receives the contributions of this call's anchors: callers that split a batch over several calls / ranks sum the dw vectors. */
int dae_triplet_batch_hard_rows(const float* D_slabs, int32_t d_splits, int64_t slab_stride, int64_t ldd,
                                const int32_t* labels, int32_t B, int32_t Bp, int32_t a0, int32_t n_anchors,
                                float* loss_part, uint32_t* cnt_part, int32_t* dw, float* G, void* stream);

/* Reduce miner partials into the normalisers / statistics:
 *   tri_scalars[0] = alpha / (N + 1e-16)  (N = N_valid | N_pos | sum cnt) -> scale of (G+G^T)
 *   tri_scalars[1] = triplet loss, [2] = fraction, [3] = num
 * For batch_hard (and pos_only) also converts the integer data_weight into dw_f32_out and cw. */
int dae_triplet_finalize(int32_t triplet, int32_t pos_only, int32_t B, int32_t Bp, float alpha,
                         const float* loss_part, const uint32_t* cnt_part, const int64_t* nvalid,
                         const int32_t* dw_i32, const uint32_t* role_cnt, float* dw_f32_out,
                         float* cw, float* tri_scalars, void* stream);

/* Gs = tri_scalars[0] * (G + G^T) on [0,B)^2 (zero elsewhere) in `dtype` [Bp x Bp] -- the A operand
 * of the triplet term of dL/dh = delta2 W + alpha (G + G^T) h (autodiff of triplet_loss_utils.py:93). */
int dae_sym_scale(const float* G, int32_t B, int32_t Bp, const float* tri_scalars, int32_t dtype,
                  void* Gs, void* stream);

/* K8 (middle): dh = sum_s slab_s (+ dh_extra);  delta1 = dh * act'(z1);  delta1^T in `dtype`
 * [Hp x ldt]; partial column sums for db_h = sum_i delta1 - act'(bh) * sum_i dh (the -act(bh) term
 * of autoencoder.py:389).  colsum_part: [2 x (Bp/32) x Hp].  delta1_f32 optional [Bp x ldh]. */
int dae_dh_finish(const float* slabs, int32_t splits, int64_t slab_stride, int64_t ld_slab,
                  const float* dh_extra, const float* h_f32, int64_t ldh, const float* bh,
                  int32_t B, int32_t H, int32_t enc_act, int32_t dtype, void* delta1_t, int64_t ldt,
                  float* colsum_part, float* delta1_f32, void* stream);

/* Reduce the bias-gradient partials into the flat gradient buffer [dW | dbh | dbv]. */
int dae_bias_grads(const float* dbv_part, int32_t n_row_waves, const float* colsum_part, int32_t n_row_blocks,
                   float* bh, int32_t H, int32_t Hp, int32_t F, int32_t Fp, int32_t enc_act,
                   float* dbh, float* dbv,
                   /* apply != 0: also run the optimizer update of bh / bv here (slots laid out [bh | bv]) */
                   int32_t apply, int32_t opt, float lr, float momentum, float grad_scale, float* bv,
                   float* s1_bias, float* s2_bias, void* stream);

/* K9: optimizer step on the padded flat parameter vector [W (Fp*Hp) | bh (Hp) | bv (Fp)] (grad, s1,
 * s2 share that layout), refreshing the low-precision shadows W_lo [Fp x Hp] and W^T_lo [Hp x Fp]
 * (autoencoder.py:444-477; tf.train.* semantics: Adagrad accumulator starts at 0.1, Momentum without
 * Nesterov, Adam with lr = lr_t precomputed by the caller).  grad_scale multiplies the gradient first
 * (1/world_size for data parallel).  apply: 0 only refreshes the shadows, 1 updates W and the biases,
 * 2 updates W only (the biases were updated by dae_bias_grads). */
int dae_opt_step(int32_t opt, float lr, float momentum, float grad_scale,
                 float* W, float* bh, float* bv, const float* grad, float* s1, float* s2,
                 int32_t Fp, int32_t Hp, int32_t dtype, void* W_lo, void* Wt_lo, int32_t apply, void* stream);

/* Final per-step statistics (autoencoder.py:233 fetch list):
 * ae = sum_i cw_i * rowloss_i (from rowloss_part + cw, or from the decode kernel's tile_part if given);
 * cost = ae + alpha * triplet.  stats: float[DAE_STATS_STRIDE]. */
/* Pieces of the data-parallel (sharded-optimizer) second half, see dae_plan_apply_rows: the optimizer on rows [f0, f1) of W
 * (W_lo rows refreshed, Wt_lo untouched), the bias update alone, and Wt_lo = W_lo^T. */
int dae_opt_step_rows(int32_t opt, float lr, float momentum, float grad_scale, float* W, const float* grad_rows, float* s1,
                      float* s2, int32_t Hp, int32_t f0, int32_t f1, int32_t dtype, void* W_lo, void* stream);
int dae_opt_bias(int32_t opt, float lr, float momentum, float grad_scale, float* bh, float* bv, const float* grad_b, float* s1b,
                 float* s2b, int32_t Hp, int32_t Fp, void* stream);
int dae_transpose_shadow(const void* W_lo, int32_t Fp, int32_t Hp, int32_t dtype, void* Wt_lo, void* stream);

int dae_step_stats(const float* rowloss_part, int32_t n_col_waves, const float* tile_part, int32_t n_tiles,
                   const float* cw, int32_t B, int32_t Bp, int32_t triplet, float alpha,
                   float* tri_scalars, const int64_t* nvalid,
                   /* optional batch_all miner partials: reduces them here instead of dae_triplet_finalize */
                   const float* loss_part, const uint32_t* cnt_part, float* stats, void* stream);

/* Explicit (anchor,pos,neg) triplet term of DenoisingAutoencoderTriplet
 * (autoencoder_triplet.py:308-311): t_i = h_i.hneg_i - h_i.hpos_i; loss = mean softplus(t).
 * h3 = [org; pos; neg] stacked compactly (3*B rows, stride ldh); writes alpha * dloss/dh3 into dh3
 * (same layout), loss_part[B] and tri_scalars[1] = loss. */
int dae_explicit_triplet(const float* h3, int64_t ldh, int32_t B, int32_t H, float alpha,
                         float* dh3, float* loss_part, float* tri_scalars, void* stream);

/* Stand-alone per-row reconstruction loss of a given decode (triplet_loss_utils.py:268-273) for the
 * function-level weighted_loss() API: x, y fp32 [B x F]; rowloss[B].  The weighted mean
 * sum(row*w)/(sum w + 1e-16) (:275) is then taken by dae_step_stats with n_col_waves = 1. */
int dae_weighted_loss_rows(const float* x, int64_t ldx, const float* y, int64_t ldy, int32_t B, int32_t F,
                           int32_t loss_func, float* rowloss, void* stream);

/* A/B switch for the plain GEMM's staging: 0 = register staging (2 LDS buffers), 2/3/4 = depth of the
 * global_load_lds ring with counted vmcnt waits (default 2); -1 / -2 = 4-wave kernel for every grid / 8-wave
 * producer-consumer kernel for grids of at most one workgroup per CU (default); -3 / -4 / -5 = dW + optimizer on the 4-wave
 * 128 x 128 kernel / on the 8-wave 160 x 128 kernel when its grid fills one round of the chip (default) / whenever it fits. */
void dae_set_glds(int32_t nst);

/* K slices with which dae_gemm_nt runs a bf16 shape on the 256 x 256-tile kernel (8 MFMA waves, one workgroup per CU; the large
 * split-K contractions of the dense-input configs: x~[B x F].W and delta2.W with F = 50 000) -- calling dae_gemm_nt with exactly this
 * `splits` selects it; 0 = the shape stays on the 128 x 128 kernels.  dae_set_glds(-6) / (-7) turn the kernel off / on (A/B). */
int32_t dae_gemm_w8_splits(int32_t dtype, int32_t M, int32_t N, int32_t K);

/* ---------------------------------------------------------------------------------------------
 * Whole-step driver: what DenoisingAutoencoder._run_train_step (autoencoder.py:206-246) does per
 * mini-batch, as one host call that enqueues every kernel above on `stream`.
 * ------------------------------------------------------------------------------------------- */
typedef struct dae_plan dae_plan;

typedef struct {
    int32_t n_features, n_components, max_batch;
    int32_t dtype, enc_act, dec_act, loss_func, opt, triplet;
    int32_t pos_triplets_only;
    int32_t encode_splits, dh_splits, gram_splits;   /* 0 = pick automatically */
    float   learning_rate, momentum, alpha;
} dae_config;

typedef struct {
    /* train set, CSR (sorted column ids) or dense fp32; exactly one of indptr / dense non-NULL */
    const int64_t* indptr; const int32_t* indices; const float* values;
    const float* dense; int64_t ld_dense;
    int64_t n_rows, nnz;
    /* parameters, padded: W [Fp x Hp], bh [Hp], bv [Fp]; flat gradient [Fp*Hp + Hp + Fp];
     * optimizer slots (same length as the gradient; may be NULL for SGD) */
    float* W; float* bh; float* bv; float* grad; float* opt_s1; float* opt_s2;
    /* low-precision shadows W_lo [Fp x Hp], Wt_lo [Hp x Fp] in cfg.dtype */
    void* W_lo; void* Wt_lo;
    void* workspace; uint64_t workspace_bytes;
    /* optional (may be NULL): bf16 image [Fp x Hp] of the W gradient.  When set, phase 1 / 5 steps in bf16 mode write the W part of the
     * gradient THERE instead of into `grad` (the bias parts stay in `grad`): the operand of a bf16 reduce-scatter, produced by the dW
     * kernel's epilogue instead of a separate cast pass. */
    void* grad_lo;
} dae_buffers;

typedef struct {
    const int32_t* row_idx;      /* device int32[B]: rows of the train set in this batch */
    const int32_t* labels;       /* device int32[B] (NULL for triplet none) */
    int32_t B;
    int32_t corr_mode; const uint32_t* keep_bits; uint64_t seed; uint32_t rng_stream;
    float corr_frac, scale;
    /* optional second CSR holding an already-corrupted copy of the train set (salt&pepper etc.) */
    const int64_t* c_indptr; const int32_t* c_indices; const float* c_values;
    const int32_t* c_row_idx;    /* rows of the corrupted CSR to take (NULL: the same row_idx as the clean set) */
    float* stats;                /* device float[DAE_STATS_STRIDE] for this step */
    int32_t phase;               /* 0 = forward+backward+update, 1 = forward+backward only (DP:
                                    caller all-reduces `grad` then calls dae_plan_apply), 2 = forward only,
                                    3 = as 0 but the W part of `grad` is not materialised (bf16: the optimizer
                                    runs in the dW GEMM's epilogue and nothing reads the gradient image),
                                    4 / 5 = the step split around an EXTERNAL miner (data parallel, global-batch mining):
                                    4 stops after the encode (h_f32, h_lo, h_t and the side images stay in the workspace);
                                    the caller mines over the all-gathered batch and writes the plan buffers `cw` (row
                                    weights), `tri_scalars`[1..3] (triplet loss, fraction, num) and `dh_extra`
                                    (d alpha*triplet / dh of its rows); 5 resumes at the decode with the same row_idx
                                    and ends like 1 */
    int32_t adam_t; float grad_scale;
} dae_step;

/* -------------------------------------------------------------------------------------------------
 * Evaluation step after the training path: N x N similarity of row vectors.
 * Replaces helpers.pairwise_similarity (helpers.py:11-50; called by main_autoencoder.py:307-317):
 *   [sklearn.preprocessing.normalize(X, norm)]  ->  cosine_similarity | linear_kernel  ->  fill_diagonal(0).
 *   X [N x ldx] fp32 (device), norm: 0 none / 1 'l1' / 2 'l2' / 3 'max', metric: 0 'cosine' / 1 'linear kernel'
 *   (any other metric is rejected, as the reference's assert does), zero_diagonal as set_diagonal_zero.
 *   out: fp32 image [dae_pad(N) x ldo], ldo >= dae_pad(N); rows / columns >= N are zero.
 *   workspace: dae_pairwise_similarity_workspace(N, D) bytes, 16-byte aligned (the normalised operand image).
 * Exact-fp32 MFMA product, fp32 accumulation.
 * ------------------------------------------------------------------------------------------------- */
uint64_t dae_pairwise_similarity_workspace(int32_t N, int32_t D);
int dae_pairwise_similarity(const float* X, int64_t ldx, int32_t N, int32_t D, int32_t norm, int32_t metric,
                            int32_t zero_diagonal, float* out, int64_t ldo, void* workspace,
                            uint64_t workspace_bytes, void* stream);

/* Related / unrelated pair statistics of an N x N similarity matrix: the numbers behind
 * helpers.visualize_pairwise_similarity (helpers.py:79-135) -- AUROC of "same label" vs "different label" over the strict
 * lower triangle (labels < 0 are missing and drop their pairs; ties count half, as sklearn's roc_curve + auc do) and the
 * box-plot statistics of the two score populations.
 *   S [N x lds] fp32 on the device; labels_host int32[N] on the HOST (the class sizes come from its histogram);
 *   out16 (host): [0] auroc, [1] n_related, [2] n_unrelated, [3] mean related, [4] mean unrelated,
 *                 [5..9] related min, q1, median, q3, max, [10..14] unrelated min, q1, median, q3, max (numpy 'linear'
 *                 percentiles); NaN where a population is empty.
 * Synchronises the stream (returns host numbers).  workspace: dae_pair_stats_workspace(N) bytes, 256-byte aligned. */
uint64_t dae_pair_stats_workspace(int32_t N);
int dae_pair_stats(const float* S, int64_t lds, const int32_t* labels_host, int32_t N, double* out16, void* workspace,
                   uint64_t workspace_bytes, void* stream);

int      dae_plan_create(const dae_config* cfg, dae_plan** out);
void     dae_plan_destroy(dae_plan* p);
uint64_t dae_plan_workspace_bytes(const dae_plan* p);
int      dae_plan_bind(dae_plan* p, const dae_buffers* bufs);
/* refresh W_lo / Wt_lo from W (after set_params / checkpoint restore) */
int      dae_plan_sync_shadows(dae_plan* p, void* stream);
/* Code-path choice of a plan, for A/B measurements and equivalence tests (every option selects between implementations of the
 * same arithmetic; the library never reads the environment).  Names: "encode_sparse" (CSR inputs: fused corrupt + gather + encode on the stored
 * entries, dae_encode_csr -- default on; 0 = dense MFMA encode GEMM), "encode_bits" (dense path: x~ as a bit image into the encode
 * GEMM; on by default for binary CSR + bf16), "x_bits" (clean rows as a bit image into the decode epilogue), "fused_opt" (optimizer in the
 * dW GEMM's epilogue), "tail" (bias gradients + statistics + x~^T un-scatter in one launch), "label_with_encode", "ce_literal"
 * (cross_entropy always by the reference-literal formula), "overlap" (batch_all: the decode kernel forks onto a side stream beside the Gram -> miner chain and joins before the dh
 * GEMM; off: measured slower, two cross-stream waits per step), "gather_tile" (process-wide: tile of the dense-ndarray gather, bit 0 = 128
 * features instead of 64, bit 1 = 128 rows instead of 64; 0 is the measured best), "miner_tile" (process-wide: 1 = lane-grid batch_all kernel,
 * default), "miner_order" / "miner_ranges" / "sym_in_decode" (side jobs riding on other launches), "gram_fp32" (exact-fp32 Gram
 * matrix in bf16 mode; before dae_plan_bind only), "dw_bits" (binary CSR + bf16: x~^T reaches the dW kernel as a bit image and the A tiles of
 * its x~^T.delta1 segment are built in LDS instead of streamed -- off by default: measured slower than the dense image), "miner_pack"
 * (batch_all workgroups = one resident round, each walking the anchor list in snake order; 0 = one workgroup per anchor), "encode_w32" (bf16 mode: the
 * sparse encode reads the fp32 master weights, so h -- and with the split-bf16 Gram matrix the triplet leg -- is fp32-accurate; default
 * on; a sharded-optimizer exchange must turn it off because only W_lo is current on every rank), "encode_w32_cols" (128 | 64 columns
 * per workgroup of that kernel), "x3_dec_wlo" / "x3_dh_hlo" (split-bf16 mode, before dae_plan_bind only, default 1: the decode multiplies
 * (h_hi, W_lo) and the dh GEMM (Gs, h^T_lo) too; 0 drops the term -- a CPU replay of the 20-step curve called both droppable, the GPU run against
 * the frozen reference curve then measured cost 7.0e-5 / triplet 1.56e-4, outside the 1e-4 gate, so they stay on: profiles/r04_precision_terms.txt;
 * with x3_dec_wlo = 0 the dW epilogue skips the lo image of the row-major shadow, which only that decode term reads), "dw_pair" (split-bf16 mode: the dW kernel runs the K segments that share their A operand
 * -- x~^T . [delta1^T_hi ; delta1^T_lo], delta2^T_hi . [h^T_hi ; h^T_lo] -- as paired ring stages of one A tile and two B tiles; default 1, 0 = one
 * segment after the other), "x3_terms" (split mode, before dae_plan_bind only: bit mask of the lo product terms that are multiplied -- bit 0 decode
 * (h_hi, W_lo), 1 decode (h_lo, W_hi), 2 dh (delta2_hi, W^T_lo), 3 dh (delta2_lo, W^T_hi), 4 dh (Gs, h^T_lo), 5 dW (x~^T, delta1^T_lo), 6 dW (delta2^T_hi,
 * h^T_lo), 7 dW (delta2^T_lo, h^T_hi), 8 / 9 dense-input encode (x~_hi, W^T_lo) / (x~_lo, W^T_hi), 10 lo images of valued inputs (clean rows, x~^T); default
 * all (bf16 storage) or bits 0 and 2 (fp16 storage); a lo image whose terms are all off is neither written nor allocated), "op_scale_log2" (16-bit modes:
 * the images of delta2, delta2^T, Gs and delta1^T hold 2^value times the quantity and the consuming epilogues divide it out; default 0 for bf16 storage,
 * log2 of the largest power of two <= 16 * max_batch (at most 14) for fp16 storage, whose normal range ends at 6.1e-5), "gram64" (16-bit modes, before
 * dae_plan_bind: the split Gram matrix on 64 x 64 tiles over the whole K, ONE slab -- default 1; 0 = 128 x 128 tiles, split-K slabs summed by the miner),
 * "decode_bn" (16-bit modes, before dae_plan_bind: tile width of the decode kernel, 64 | 128 | 0 = by tile count: 128 once the 64-column tiles are more than
 * four rounds of the chip, i.e. F = 50000), "dw_tr" (-1 | 0 | 1: the dW kernel reads x~ and delta2 row-major through transposing LDS reads, so that
 * delta2^T / x~^T are never written; -1 = for dense train sets only, where it was measured faster), "dw_rounds" (process-wide: rounds of the chip the fused
 * 160 x 128 dW + optimizer kernel may take, default 16 for the split modes / 1 otherwise), "pad_skip" (process-wide, default 1: the decode epilogue does not
 * evaluate the loss of 32-row blocks that are pure batch padding).  Unknown names are an error. */
int      dae_plan_set_option(dae_plan* p, const char* name, int32_t value);
/* 16-bit storage format the loaded library was built for: 0 = bfloat16 (libdae_hip.so), 1 = IEEE fp16 (libdae_hip_f16.so: the same sources compiled with
 * -DDAE_F16=1; every 16-bit image and the MFMA that multiplies it switch together).  DAE_BF16 / DAE_BF16X3 name "the 16-bit format" in either build. */
int32_t  dae_storage_format(void);
int      dae_train_step(dae_plan* p, const dae_step* step, void* stream);
int      dae_plan_apply(dae_plan* p, int32_t adam_t, float grad_scale, void* stream);
/* dae_plan_apply restricted to the rows [f0, f1) of W (multiples of 64; every low-precision image of those rows is rebuilt): the bucketed form of the
 * data-parallel all-reduce applies band k while band k + 1 is still being reduced.  The band with f1 == Fp also updates the biases. */
int      dae_plan_apply_band(dae_plan* p, int32_t adam_t, float grad_scale, int32_t f0, int32_t f1, void* stream);
/* Data parallel with a SHARDED optimizer: after dae_train_step(phase = 1) the ranks reduce-scatter the W part of the flat gradient
 * by row chunks and all-reduce its (small) bias part; every rank then updates the rows [f0, f1) it owns from grad_rows (fp32
 * [f1-f0 x Hp], rank-summed) and -- update_bias != 0 -- the biases, all-gathers W_lo and rebuilds Wt_lo = W_lo^T locally
 * (dae_plan_refresh_wt).  Per step and rank this moves 1/2 (fp32 gradients) to 1/4 (bf16 gradients) of an fp32 all-reduce. */
int      dae_plan_apply_rows(dae_plan* p, int32_t adam_t, float grad_scale, const float* grad_rows, int32_t f0, int32_t f1,
                             int32_t update_bias, void* stream);
int      dae_plan_refresh_wt(dae_plan* p, void* stream);

/* Packed exchange of the data-parallel step (two collectives per step instead of three, no copy, one rebuild kernel):
 *   dae_plan_apply_rows_packed: the sharded optimizer step of dae_plan_apply_rows, but the low-precision rows of [f0, f1) are written
 *     into the all-gather SEND buffer (row f at send + (f - f0) * Hp * elem) and this rank's LOCAL bias gradients (Hp + Fp floats) are
 *     copied to send + bias_off_bytes; the biases are not updated.
 *   dae_plan_dp_unpack: after all-gathering the send buffers into recv (world chunks of chunk_stride_bytes): W_lo <- the rows,
 *     Wt_lo <- their transpose, and the biases are updated from the rank-ordered sum of the gathered bias gradients (every rank
 *     computes the same sum).  dae_dp_unpack is the plan-free form. */
/* Data parallel: `stream` waits until the W gradient of the last enqueued dae_train_step is complete (an event between the dW GEMM and
 * the step's tail kernel), so that a reduce-scatter issued on `stream` runs beside the tail.  The first call only creates the event and
 * returns DAE_WAIT_DW_CREATED (not an error, dae_last_error untouched; steps enqueued earlier are not covered: wait for the step's
 * stream instead); 0 = the wait was enqueued; any other value is an error. */
#define DAE_WAIT_DW_CREATED 100   /* outside the error codes (1 = bad argument, 2 = HIP failure) */
int dae_plan_stream_wait_dw(dae_plan* plan, void* stream);
int dae_plan_apply_rows_packed(dae_plan* plan, int32_t adam_t, float grad_scale, const float* grad_rows, int32_t f0, int32_t f1,
                               void* send, int64_t bias_off_bytes, void* stream);
int dae_plan_dp_unpack(dae_plan* plan, const void* recv, int32_t world, int32_t chunk_rows, int64_t chunk_stride_bytes,
                       int64_t bias_off_bytes, int32_t adam_t, float grad_scale, void* stream);
int dae_dp_unpack(const void* recv, int32_t world, int32_t chunk_rows, int64_t chunk_stride_bytes, int64_t bias_off_bytes,
                  int32_t Fp, int32_t Hp, int32_t dtype, void* W_lo, void* Wt_lo, int32_t opt, float lr, float momentum,
                  float grad_scale, float* bh, float* bv, float* s1b, float* s2b, float* grad_b, void* stream);
/* -------------------------------------------------------------------------------------------------
 * The data-parallel collective in the C ABI (SURVEY 8(b) proposed `dae_allreduce_grads`; SURVEY 8(e): all-reduce(sum) of the flat gradient
 * [dW | dbh | dbv] over xGMI, then the identical optimizer step on every rank).  Shards the reference's per-mini-batch step,
 * autoencoder/autoencoder.py:206-246 (one session.run), over one process per GPU.
 *
 *   dae_comm_unique_id   rank 0 draws a DAE_COMM_ID_BYTES id (ncclGetUniqueId) and shares it with the other ranks over any out-of-band
 *                        channel the host has (a file, a socket, torch.distributed's store, MPI).
 *   dae_comm_init        every rank, on its own device (the calling thread's current HIP device): ncclCommInitRank.  world = 1 is valid (a
 *                        one-rank communicator: the exchange then equals dae_plan_apply).  RCCL is located with dlopen at this point -- a
 *                        copy already loaded into the process is shared, else librccl.so.1 of the ROCm install -- so single-GPU users of the
 *                        library never load it; a missing RCCL is an error here, not at load time.
 *   dae_allreduce_grads  in-place ncclAllReduce(sum, fp32) of the plan's flat gradient, enqueued on `stream` -- the stream the step's kernels
 *                        run on: no process-group stream, no cross-stream hop, no host synchronisation.
 *   dae_dp_exchange      all-reduce + optimizer, i.e. everything behind dae_train_step(phase = 1).  buckets <= 1: dae_allreduce_grads +
 *                        dae_plan_apply back to back on `stream`.  buckets > 1 (at most DAE_COMM_MAX_BUCKETS): the flat buffer is reduced in
 *                        row bands of W (dae_dp_bands) on the communicator's own wire stream -- the W-only bands start behind the dW GEMM,
 *                        beside the step's tail kernel; the last band (it carries the bias gradients) behind the tail -- and `stream`
 *                        applies band k (dae_plan_apply_band) while band k + 1 is on the wire.  Same sums, same update as one bucket.
 *   dae_dp_bands         the band boundaries (host arithmetic; bounds[0 .. n], bounds[n] = Fp), returns n.
 *   dae_comm_allreduce_f32  the bare collective on a caller buffer (op 0 = sum, 1 = max): loss normalisers, "ranks seen" tokens, timings.
 * Return codes as everywhere (0 ok, 1 bad argument, 2 HIP failure) plus 3 = RCCL failure; text in dae_last_error().
 * ------------------------------------------------------------------------------------------------- */
#define DAE_COMM_ID_BYTES 128
#define DAE_COMM_MAX_BUCKETS 8
typedef struct dae_comm dae_comm;
int         dae_comm_unique_id(void* id_out);
int         dae_comm_init(const void* id, int32_t rank, int32_t world, dae_comm** out);
void        dae_comm_destroy(dae_comm* c);
/* out4 = {rank, world, RCCL version code, DAE_COMM_MAX_BUCKETS} */
int         dae_comm_info(const dae_comm* c, int32_t* out4);
/* which RCCL the library resolved ("" before the first dae_comm_unique_id / dae_comm_init) */
const char* dae_comm_library(void);
int         dae_comm_allreduce_f32(dae_comm* c, float* buf, int64_t n, int32_t op, void* stream);
int32_t     dae_dp_bands(int32_t Fp, int32_t buckets, int32_t* bounds);
int         dae_allreduce_grads(dae_plan* p, dae_comm* c, void* stream);
int         dae_dp_exchange(dae_plan* p, dae_comm* c, int32_t adam_t, float grad_scale, int32_t buckets, void* stream);

/* transform(): out[B x H] fp32 (ld_out) = encode of rows row_idx (autoencoder.py:479-505) */
int      dae_encode_rows(dae_plan* p, const int32_t* row_idx, int32_t B, float scale,
                         const int64_t* indptr, const int32_t* indices, const float* values,
                         const float* dense, int64_t ld_dense, float* out, int64_t ld_out, void* stream);
/* pointers into the workspace for tests / debugging (NULL if name unknown) */
void*    dae_plan_buffer(dae_plan* p, const char* name);
/* Per-kernel timing of dae_train_step with HIP events recorded on the step's own stream (bench.py's
 * roofline leg).  enable = 1: every launch is bracketed by two events and the host waits for it (a launch
 * starts on an idle device).  enable = 2 (queued): every launch of a step has its own event pair, nothing
 * waits between launches or steps, the pairs are read when the pool is full and by dae_plan_profile_read
 * -- the kernels run back to back as they do un-profiled and the averages come within the two markers' cost of rocprofv3's.  Throughput numbers
 * enable = 3 (kernel timestamps): like 2, but the pair of a kernel launch is handed to hipExtLaunchKernelGGL
 * and carries the dispatch's own begin / end times (no marker packets in the stream; a call with several
 * launches adds them up) -- the figure rocprofv3 --kernel-trace reports.  Throughput numbers
 * must be taken with profiling off (enable = 0).  Slot names via dae_plan_profile_name(). */
int         dae_plan_profile(dae_plan* p, int32_t enable);
int         dae_plan_profile_read(const dae_plan* p, int32_t max_slots, double* ms_total, int32_t* counts);
int32_t     dae_plan_profile_slots(void);
const char* dae_plan_profile_name(int32_t slot);
/* out8 = {Fp, Hp, Bp_max, encode_splits, dh_splits, gram_splits, element_size, 0} */
int      dae_plan_info(const dae_plan* p, int32_t* out8);

#ifdef __cplusplus
}
#endif
#endif /* DAE_HIP_H */
