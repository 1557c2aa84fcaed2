// dae_kernels.h -- internal (non-ABI) declarations shared by the translation units of libdae_hip.so.
#pragma once
#include "dae_common.h"

namespace dae {

// epilogue descriptor of the fused decode + loss kernel (dae_gemm.hip)
struct DecodeEpi {
    const float* bv;          // [Fp]
    const void* x;            // [Bp x ldx] clean input, element type T
    int64_t ldx;
    const void* x2;           // split-bf16 mode, input values not exact in bf16: lo image of x (bf16, same leading dimension); else NULL
    const uint32_t* x_bits;   // bf16 + binary input: [Bp x ldxb] bit image of x instead (bit b of word w = feature 32w+b); else NULL
    int64_t ldxb;
    const float* cw;          // [Bp] w_i / (sum w + 1e-16), zero for i >= B
    const float* cos_stats;   // cosine: [3 x Bp] = {sum x^2 | sum y^2 | sum xhat.y}; NULL otherwise
    float* rowloss_part;      // [2*tiles_n x Bp] per-row partial sums (may be NULL)
    float* tile_part;         // [tiles_m*tiles_n] sum_rows cw_i * rowloss_i of each tile (may be NULL)
    float* dbv_part;          // [2*tiles_m x Fp]
    float* cos_part;          // cosine first pass: [2 x 2*tiles_n x Bp] partial {sum y^2, sum xhat*y}
    void* delta2; int64_t ldd;
    void* delta2_t; int64_t lddt;
    void* delta2_2; void* delta2_t2;   // split-bf16 mode: lo images of delta2 / delta2^T, bf16(d - bf16(d)), same leading dimensions (NULL otherwise)
    int B, F, Bp, Fp;
    int dec_act, loss_func;
    int cos_pass;             // 0: not cosine, 1: statistics pass, 2: final pass
    int ce_literal;           // 1: always evaluate cross_entropy with the reference-literal formula (option ce_literal; A/B and tests)
    // optional rider: (sym_Bp/64)^2 extra workgroups at the end of the grid compute Gs = scalars[0] * (G + G^T) for the dh GEMM
    // (the stand-alone sym_scale launch sits between the miner and this kernel and neither depends on the other)
    const float* sym_G; const float* sym_scalars; void* sym_Gs; int sym_B, sym_Bp, sym_first;
    // 16-bit modes: delta2 / delta2^T (and the rider's Gs) are stored times this power of two (1 unless the storage format is fp16: dae_api.hip op_scale);
    // the loss, the bias-gradient partials and everything fp32 stay unscaled
    float op_scale;
    int no_pad_skip;          // 1: evaluate every 32-row block, padding included (A/B; set by the launcher from the process-wide switch)
    int dbg;                  // timing probes of gemm_decode_ast (dae_set_glds(-500000 - bits); results are garbage when set): 1 no A loads, 2 no fragment reads / MFMAs, 4 no LDS-DMA, 8 no epilogue arithmetic, 16 no tile stores
    float* z_io; int64_t ldz; // cosine only: the logits' GEMM part (accumulators, before the bias) of the statistics pass, Bp x Fp fp32 in the kernel's own register order (ldz unused) -- z_mode 1: pass 1 stores them,
    int z_mode;               // z_mode 2: pass 2 LOADS them instead of running the K loop again (the same fp32 values the recomputation would produce); 0 / NULL: recompute
    int bn;                   // tile width (columns of y per workgroup): 0 = the mode's default (decode_tile_n), 128 = the wide 16-bit kernel; the partial-sum arrays
                              // (rowloss_part / cos_part: 2 * Fp / bn rows; tile_part: (Bp / 128) * (Fp / bn)) are laid out by it
};

// lo product terms of the split 16-bit mode (dae_config.dtype = DAE_BF16X3; plan option "x3_terms"): each bit keeps one (hi, lo) / (lo, hi) product of one
// contraction -- and with it the lo image it reads.  bf16 storage needs them all (profiles/r04_precision_terms.txt); fp16 storage holds the 1e-4 curve
// with the two W terms alone (tools/precision_study.py --scheme W=f16split).
enum : uint32_t {
    X3T_DEC_WLO = 1u << 0,    // decode   (h_hi, W_lo)
    X3T_DEC_HLO = 1u << 1,    // decode   (h_lo, W_hi)
    X3T_DH_WLO = 1u << 2,     // dh       (delta2_hi, W^T_lo)
    X3T_DH_D2LO = 1u << 3,    // dh       (delta2_lo, W^T_hi)
    X3T_DH_HLO = 1u << 4,     // dh       (Gs, h^T_lo)
    X3T_DW_D1LO = 1u << 5,    // dW       (x~^T, delta1^T_lo)
    X3T_DW_HLO = 1u << 6,     // dW       (delta2^T_hi, h^T_lo)
    X3T_DW_D2LO = 1u << 7,    // dW       (delta2^T_lo, h^T_hi)
    X3T_ENC_WLO = 1u << 8,    // dense-input encode (x~_hi, W^T_lo)
    X3T_ENC_XLO = 1u << 9,    // dense-input encode (x~_lo, W^T_hi)
    X3T_XV = 1u << 10,        // valued input: lo images of the clean rows x (decode epilogue) and of x~^T (dW: (x~^T_lo, delta1^T_hi))
    X3T_ALL = (1u << 11) - 1,
    // (the dense-input encode's (x~, W^T_lo) term is NOT in the fp16 default: the 20-step curve of c4 does not move without it -- cost 7.0e-6 / triplet
    //  1.7e-5 against 7.6e-6 / 1.3e-5 with it, profiles/r05_c4_terms.txt -- and the encode GEMM at F = 50000 takes 114 instead of 191 us)
    X3T_F16_DEFAULT = X3T_DEC_WLO | X3T_DH_WLO,
};

struct LabelJob;
// one K segment of a contraction: A_seg [M x K] and Bt_seg [N x K], both K-contiguous (leading dimensions in ELEMENTS); K = 0 segments are skipped
struct GemmSegDesc { const void* A; int64_t lda; const void* Bt; int64_t ldb; int K; };
// label_job: when the launch uses the 8-wave kernel and leaves a CU free, one extra workgroup computes the label statistics
// (*label_done = 1); otherwise the caller launches them itself
int launch_gemm_f32out(int dtype, int M, int N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0, int K0,
                       const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int K1, float* C, int64_t ldc, int splits,
                       int64_t slab_stride, hipStream_t st, int role = 0, const struct LabelJob* label_job = nullptr,
                       int* label_done = nullptr, int m_valid = 0);
// the same contraction over up to 5 K segments (split-bf16 operands: (hi,hi) (hi,lo) (lo,hi) per product)
int launch_gemm_f32out_n(int dtype, int M, int N, const GemmSegDesc* segs, int nsegs, float* C, int64_t ldc, int splits, int64_t slab_stride,
                         hipStream_t st, int role = 0, const LabelJob* label_job = nullptr, int* label_done = nullptr, float out_scale = 1.f,
                         int m_valid = 0);   // m_valid: rows of A that hold data (0 = all M): the producer/consumer kernel skips the MFMAs of all-padding 32-row blocks
// (out_scale: C = out_scale * sum -- un-split launches only; the dW gradient of a scaled delta image leaves unscaled)
// K slices the 256 x 256 / 8-MFMA-wave kernel wants for this shape (0: the shape stays on the 128 x 128 kernels); see dae_gemm.hip
int gemm_w8_splits(int dtype, int M, int N, int ktiles);
enum { GEMM_ROLE_GENERIC = 0, GEMM_ROLE_ENCODE = 1, GEMM_ROLE_DH = 2, GEMM_ROLE_DW = 3, GEMM_ROLE_GRAM = 4 };
int launch_decode_loss(int dtype, int Bp, int Fp, int Hp, const void* h_lo, int64_t ldh, const void* W_lo, int64_t ldw,
                       const DecodeEpi& e, hipStream_t st);
int launch_decode_loss_n(int dtype, int Bp, int Fp, const GemmSegDesc* segs, int nsegs, const DecodeEpi& e, hipStream_t st);
// tile width (columns of y per workgroup) of the decode kernel for `dtype`: the partial-sum arrays it writes
// (rowloss_part / cos_part: 2 * Fp / width rows; tile_part: (Bp/128) * (Fp/width) entries) are laid out by it
int decode_tile_n(int dtype);
// D = hcat_a . hcat_b^T (the split 16-bit Gram operands, K = 3 Hp) on 64 x 64 tiles, one slab [Bp x Bp] (dae_gemm.hip: gram64_kernel)
int launch_gram64(const void* hcat_a, const void* hcat_b, int Bp, int Hp, float* D, hipStream_t st);
// label statistics job (dae_label.h): either its own launch or an extra block of the CSR gather kernel
struct LabelJob {
    const int32_t* labels; int B, Bp, triplet; int64_t* nvalid; int64_t* dw; float* cw; float alpha; float* tri_scalars;
    int32_t* order;           // [B] or NULL: the batch rows by DESCENDING batch_all sweep cost (n-1)(B-n), ties by index -- the
                              // dispatch order of the miner's workgroups (longest anchors first, the short ones fill the tail)
    int32_t* cls;             // [1 + 2 B] or NULL: cls[0] = 1 iff the labels are non-decreasing (a class-sorted batch); then
                              // cls[1 + 2i], cls[2 + 2i] = first index and end of row i's class (the miner's range fast path)
};
int launch_gather_csr(const int64_t* indptr, const int32_t* indices, const float* values, const int32_t* row_idx, int B, int F, int dtype,
                      void* x, void* xc, int64_t ldx, void* xct, int64_t ldt, float* rowsq, int corr_mode, const uint32_t* keep_bits,
                      uint64_t seed, uint32_t rng_stream, float corr_frac, float scale, uint32_t* xc_bits, int64_t ldw,
                      const LabelJob* label_job, hipStream_t st, uint32_t* x_bits = nullptr, void* x2 = nullptr);   // x2: lo image of x (split-bf16, needs xc == NULL)

// dae_gather_dense with res = 1: the RESIDUAL images v - bf16(v) (split-bf16 mode: lo parts of x, x~, x~^T; bf16 only, no row squares)
int launch_gather_dense(const float* data, int64_t ld_data, const int32_t* row_idx, int32_t B, int32_t F, int32_t dtype, void* x, void* xc, int64_t ldx,
                        void* xct, int64_t ldt, float* rowsq, float* rowsq_scratch, int32_t corr_mode, const uint32_t* keep_bits, uint64_t seed,
                        uint32_t rng_stream, float corr_frac, float scale, void* stream, int res);
// dae_encode_finish with the lo image of h^T (split-bf16 mode; NULL otherwise)
int launch_encode_finish(const float* slabs, int32_t splits, int64_t slab_stride, int64_t ld_slab, const float* bh, int32_t B, int32_t H, int32_t enc_act,
                         int32_t dtype, float* h_f32, void* h_lo, int64_t ldh, void* h_t, int64_t ldht, void* hcat_a, void* hcat_b, void* h_t2, void* stream);

// fused corrupt + gather + encode for CSR inputs (dae_gather.hip: encode_csr_kernel)
struct EncCsrLaunch {
    const int64_t* indptr; const int32_t* indices; const float* values; const int32_t* row_idx;
    int B, F, H, dtype;
    const void* W; int64_t ldw; const float* bh; int enc_act;
    int corr_mode; const uint32_t* keep_bits; uint64_t seed; uint32_t rng_stream; float corr_frac, scale;
    float* h_f32; void* h_lo; int64_t ldh; void* h_t; int64_t ldht; void* hcat_a; void* hcat_b;
    uint32_t* x_bits; int64_t ldxb; void* xct; int64_t ldt; float* rowsq;
    const LabelJob* label_job;
    uint32_t* xtb; int64_t ldxt;   // x~^T as a bit image [Fp x ldxt words] (binary data; pre-zeroed) instead of the dense xct
    int w_f32;                     // bf16 activations only: W points at the fp32 MASTER weights [Fp x ldw] (h is then fp32-accurate)
    int w32_cols;                  // w_f32: 64 (default: one 2.6 MB slice per XCD L2) or 128 columns per workgroup
    void* h_t2;                    // split-bf16 mode: lo image of h^T [Hp x ldht] (h_t holds hi); NULL otherwise
    void* xct2;                    // split-bf16 mode, x~ not exact in bf16: lo image of x~^T (layout of xct, pre-zeroed); NULL otherwise
    int xct_rm;                    // 1: `xct` (and xct2) is the ROW-MAJOR image x~ [Bp x ldt] instead of x~^T [Fp x ldt] -- entry (i, col) at i * ldt + col: the operand
                                   // layout of the dW kernel's transposed-A form (gemm_dw_pc<TRA>), which reads its A tiles [k = batch][m = feature]
};
int launch_encode_csr(const EncCsrLaunch& q, hipStream_t st);
size_t encode_csr_lds_bytes(int dtype, int w_f32, int w32_cols, int64_t ldxb);

// ---- argument packs of the step-tail kernel (bias gradients + statistics + x~^T un-scatter in one launch) ----
struct BiasArgs {
    const float* dbv_part; int n_row_waves; const float* colsum_part; int n_row_blocks;
    float* bh; int H, Hp, F, Fp, enc_act; float* dbh; float* dbv;
    int apply, opt; float lr, mom, gscale; float* bv; float* s1b; float* s2b;
};
struct StatsArgs {
    const float* rowloss_part; int n_col_waves; const float* tile_part; int n_tiles; const float* cw; int B, Bp, triplet;
    float alpha; float* tri_scalars; const int64_t* nvalid; float* stats; const float* loss_part; const uint32_t* cnt_part;
};
struct ClearArgs {            // CSR rows whose entries were scattered into x~^T [Fp x ldt] this step
    const int64_t* indptr; const int32_t* indices; const int32_t* row_idx; int B, F; void* xct; int64_t ldt; int es;
    uint32_t* xtb; int64_t ldxt;   // the bit image of x~^T instead of the dense one (xct == NULL): clears the word holding bit (i, col)
    void* xct2;                    // split-bf16 mode with inexact x~: the lo image of x~^T, cleared alongside (bf16; NULL otherwise)
    int rm;                        // 1: xct / xct2 are row-major x~ [Bp x ldt] (entry (i, col) at i * ldt + col), see EncCsrLaunch::xct_rm
};
// K8 (middle), see dae_dh_finish; delta1_lo: optional ROW-MAJOR delta1 [Bp x ldh] in `dtype` (operand of the sparse x~^T.delta1)
// K9 on the whole of W (+ biases): dae_opt_step with the lo images of the split-bf16 shadows (NULL outside that mode)
int launch_opt_step(int opt, float lr, float momentum, float grad_scale, float* W, float* bh, float* bv, const float* grad, float* s1, float* s2,
                    int Fp, int Hp, int dtype, void* W_lo, void* Wt_lo, void* W_lo2, void* Wt_lo2, int apply, void* stream, int f0 = 0, int f1 = -1);
int launch_dh_finish(const float* slabs, int splits, int64_t slab_stride, int64_t ld_slab, const float* dh_extra, const float* h_f32,
                     int64_t ldh, const float* bh, int B, int H, int enc_act, int dtype, void* delta1_t, int64_t ldt, float* colsum_part,
                     float* delta1_f32, void* delta1_lo, hipStream_t st, void* delta1_t2 = nullptr,   // delta1_t2: lo image of delta1^T (split-bf16)
                     float in_scale = 1.f, float out_scale = 1.f);   // dh = in_scale * sum(slabs) + dh_extra; the 16-bit delta1 images hold out_scale * delta1
// Gs = mul * tri_scalars[0] * (G + G^T): dae_sym_scale with the 16-bit modes' operand scale
int launch_sym_scale(const float* G, int B, int Bp, const float* tri_scalars, int dtype, void* Gs, float mul, hipStream_t st);
int launch_cast_bf16(const float* src, void* dst_bf16, int64_t n, hipStream_t st);
int launch_step_tail(const BiasArgs& ba, const StatsArgs* sa, const ClearArgs* ca, hipStream_t st);
// batch_all miner with an optional dispatch order of the anchors (dae_triplet.hip)
int launch_batch_all(const float* D_slabs, int d_splits, int64_t slab_stride, int64_t ldd, const int32_t* labels, int B, int Bp, int a0,
                     int n_anchors, int mode, float* loss_part, uint32_t* npos_part, float* G, uint32_t* role_cnt, const int32_t* order,
                     hipStream_t st, const int32_t* cls = nullptr);

void set_miner_pack(int on);
void set_miner_tile(int on);

// epilogue of the fused dW + optimizer GEMM (gemm_dw_opt): parameters updated in place from the gradient tile
struct OptEpi {
    float* W;                 // [Fp x ldw] fp32 master weights
    float* grad;              // [Fp x ldw] gradient image, or NULL when nobody reads it
    float *s1, *s2;           // optimizer slots (same layout as W), NULL when unused by `opt`
    void *W_lo, *Wt_lo;       // bf16 shadows [Fp x ldw] and [Hp x ldwt]
    void *W_lo2, *Wt_lo2;     // split-bf16 mode: their lo images, bf16(W - bf16(W)) (NULL otherwise)
    int64_t ldw, ldwt;
    int opt;                  // DAE_OPT_* or DW_OPT_GRAD_ONLY (gradient to memory, no update: `grad` fp32 and / or `grad_lo` bf16)
    float lr, mom, gscale;
    void* grad_lo;            // DW_OPT_GRAD_ONLY: bf16 gradient image [Fp x ldw] (the reduce-scatter operand of data parallel) or NULL
    float gin;                // the accumulated tile is multiplied by this first (1 / op_scale of the 16-bit delta images; 0 is read as 1)
};
enum { DW_OPT_GRAD_ONLY = 4 };
// x~^T handed to the dW kernel as a BIT image (binary CSR input): the A tiles of the x~^T.delta1 segment are built in LDS
struct DwBitsArgs {
    const uint32_t* xtb; int64_t ldxt;   // [Fp x ldxt words]: bit i of row f <=> entry (i, f) of the batch kept
    float scale;                         // value of a kept entry
};
bool dw_bits_fits(int M, int N, int Bp);
bool dw_x3_fits(int M, int N, int Bp);       // split-bf16 mode: can launch_dw_opt_n (one 160 x 128 tile per CU, whole 64-deep K tiles) run the shape?
// xa != NULL: segment 0 is x~^T (bit image, A0 ignored) . Bt0 = delta1^T; segment 1 = delta2^T . h^T as usual
int launch_dw_opt(int M, int N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0, int K0, const void* A1, int64_t lda1,
                  const void* Bt1, int64_t ldb1, int K1, const OptEpi& e, hipStream_t st, const DwBitsArgs* xa = nullptr, bool tra = false);
// tra: the A operands are ROW-MAJOR batch images [K x M] (x~, delta2: lda = their leading dimension) read through transposing LDS reads (gemm_dw_pc<TRA>)
bool dw_pc_taken(int M, int N, int K0, int K1, bool grad_only);
// split-bf16 mode (e.Wt_lo2 set, e.W_lo2 optional); pair: segments that share their A operand run as paired stages (one A tile, two B tiles)
int launch_dw_opt_n(int M, int N, const GemmSegDesc* segs, int nsegs, const OptEpi& e, hipStream_t st, bool pair = false, bool tra = false);
void set_use_glds(int nst);
void set_gather_tile(int v);          // dense gather tile: bit 0 = 128 features (else 64), bit 1 = 128 rows (else 64)
int launch_gemm_trace(int dtype, int M, int N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0, int K0, const void* A1,
                      int64_t lda1, const void* Bt1, int64_t ldb1, int K1, float* C, int64_t ldc, int splits, int64_t slab_stride,
                      int nst, unsigned long long* trace, hipStream_t st);
int launch_encode_bits(int Bp, int Hp, int Fp, const uint32_t* bits, int64_t ldw, const void* Wt_lo, int64_t ldb, float* C,
                       int64_t ldc, int splits, int64_t slab_stride, hipStream_t st, const LabelJob* label_job = nullptr,
                       int* label_done = nullptr);
bool encode_bits_fits(int Bp, int Hp, int Fp, int splits);

}  // namespace dae
