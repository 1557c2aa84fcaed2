// dae_api.hip -- extern "C" surface of libdae_hip.so (declared in include/dae_hip.h) and the whole-step
// driver that enqueues one DAE training step (DenoisingAutoencoder._run_train_step's per-batch body,
// autoencoder.py:223-245) as a fixed sequence of HIP kernels on one stream, with no host sync.
#include <stdarg.h>
#include <new>

#include "dae_kernels.h"

namespace dae {

static thread_local char g_err[1024] = "";
thread_local LaunchTimer g_lt = {nullptr, nullptr, nullptr, 0, 0, false};
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace dae

using namespace dae;

extern "C" int dae_abi_version(void) { return DAE_ABI_VERSION; }
extern "C" const char* dae_last_error(void) { return g_err; }
extern "C" int64_t dae_pad(int64_t n) { return pad128(n); }
extern "C" void dae_set_glds(int32_t nst) { set_use_glds(nst); }
extern "C" int32_t dae_gemm_w8_splits(int32_t dtype, int32_t M, int32_t N, int32_t K) { return gemm_w8_splits(dtype, M, N, K * (dtype == DAE_BF16 ? 2 : 4) / 128); }
extern "C" int32_t dae_decode_tile_n(int32_t dtype) { return decode_tile_n(dtype); }
// 16-bit storage format this library was built for: 0 = bfloat16 (libdae_hip.so), 1 = IEEE fp16 (libdae_hip_f16.so); see dae_common.h
extern "C" int32_t dae_storage_format(void) { return kF16 ? 1 : 0; }

extern "C" int dae_gemm_nt(int32_t dtype, int32_t M, int32_t N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0,
                           int32_t K0, const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int32_t K1, float* C,
                           int64_t ldc, int32_t splits, int64_t slab_stride, void* stream) {
    return launch_gemm_f32out(dtype, M, N, A0, lda0, Bt0, ldb0, K0, A1, lda1, Bt1, ldb1, K1, C, ldc, splits, slab_stride,
                              (hipStream_t)stream);
}

extern "C" int dae_gemm_nt_n(int32_t dtype, int32_t M, int32_t N, const dae_gemm_seg* segs, int32_t nsegs, float* C, int64_t ldc,
                             int32_t splits, int64_t slab_stride, void* stream) {
    DAE_CHECK_ARG(segs && nsegs >= 1 && nsegs <= 6, "gemm_nt_n: 1..6 K segments");
    GemmSegDesc d[6];
    for (int i = 0; i < nsegs; ++i) d[i] = {segs[i].A, segs[i].lda, segs[i].Bt, segs[i].ldb, segs[i].K};
    return launch_gemm_f32out_n(dtype, M, N, d, nsegs, C, ldc, splits, slab_stride, (hipStream_t)stream);
}

extern "C" int dae_gemm_trace(int32_t dtype, int32_t M, int32_t N, const void* A0, int64_t lda0, const void* Bt0, int64_t ldb0,
                              int32_t K0, const void* A1, int64_t lda1, const void* Bt1, int64_t ldb1, int32_t K1, float* C,
                              int64_t ldc, int32_t splits, int64_t slab_stride, int32_t nst, uint64_t* trace, void* stream) {
    return launch_gemm_trace(dtype, M, N, A0, lda0, Bt0, ldb0, K0, A1, lda1, Bt1, ldb1, K1, C, ldc, splits, slab_stride, nst,
                             (unsigned long long*)trace, (hipStream_t)stream);
}

extern "C" int dae_encode_bits(const uint32_t* xc_bits, int64_t ldw, const void* Wt_lo, int64_t ldwt, int32_t Bp, int32_t Hp, int32_t Fp,
                               float* slabs, int64_t ld_slab, int32_t splits, int64_t slab_stride, void* stream) {
    return launch_encode_bits(Bp, Hp, Fp, xc_bits, ldw, Wt_lo, ldwt, slabs, ld_slab, splits, slab_stride, (hipStream_t)stream);
}

extern "C" int dae_gram(const float* h_f32, int64_t ldh, int32_t Bp, int32_t Hp, float* D_slabs, int32_t splits, void* stream) {
    return launch_gemm_f32out(DAE_F32, Bp, Bp, h_f32, ldh, h_f32, ldh, Hp, nullptr, 0, nullptr, 0, 0, D_slabs, Bp, splits,
                              (int64_t)Bp * Bp, (hipStream_t)stream);
}

extern "C" int dae_decode_loss(int32_t dtype, int32_t B, int32_t F, int32_t H, const void* h_lo, int64_t ldh, const void* W_lo,
                               int64_t ldw, const float* bv, const void* x, int64_t ldx, const float* cw, int32_t dec_act,
                               int32_t loss_func, int32_t cos_pass, const float* cos_stats, float* cos_part,
                               float* rowloss_part, float* tile_part, float* dbv_part, void* delta2, int64_t ldd,
                               void* delta2_t, int64_t lddt, void* stream) {
    DAE_CHECK_ARG(h_lo && W_lo && bv && x && cw, "decode_loss: null input");
    DAE_CHECK_ARG(B > 0 && F > 0 && H > 0, "decode_loss: bad shape");
    DAE_CHECK_ARG(loss_func >= DAE_LOSS_CROSS_ENTROPY && loss_func <= DAE_LOSS_COSINE, "decode_loss: unknown loss %d", loss_func);
    if (loss_func == DAE_LOSS_COSINE) {
        DAE_CHECK_ARG(cos_pass == 1 || cos_pass == 2, "decode_loss: cosine needs cos_pass 1 or 2");
        DAE_CHECK_ARG(cos_stats && (cos_pass != 1 || cos_part), "decode_loss: cosine statistics buffers required");
    } else {
        DAE_CHECK_ARG(cos_pass == 0 && (rowloss_part || tile_part), "decode_loss: rowloss_part or tile_part required");
    }
    DecodeEpi e;
    memset(&e, 0, sizeof(e));
    e.bv = bv; e.x = x; e.ldx = ldx; e.cw = cw; e.cos_stats = cos_stats; e.rowloss_part = rowloss_part; e.tile_part = tile_part; e.dbv_part = dbv_part;
    e.cos_part = cos_part; e.delta2 = delta2; e.ldd = ldd; e.delta2_t = delta2_t; e.lddt = lddt;
    e.B = B; e.F = F; e.Bp = (int)pad128(B); e.Fp = (int)pad128(F); e.dec_act = dec_act; e.loss_func = loss_func;
    e.cos_pass = cos_pass;
    return launch_decode_loss(dtype, e.Bp, e.Fp, (int)pad128(H), h_lo, ldh, W_lo, ldw, e, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// plan: sizes, workspace carving, step driver
// ------------------------------------------------------------------------------------------------
// per-kernel HIP-event timing slots of the step driver (bench.py's roofline leg)
enum { PS_MEMSET = 0, PS_GATHER, PS_ENC_GEMM, PS_ENC_FIN, PS_LABEL, PS_GRAM, PS_MINER, PS_TRI_FIN, PS_SYM, PS_DECODE,
       PS_COS_REDUCE, PS_STATS, PS_DH_GEMM, PS_DH_FIN, PS_DW_GEMM, PS_BIAS, PS_OPT, PS_COUNT };
static const char* const kProfNames[PS_COUNT] = {"memset_xct", "gather", "encode_gemm", "encode_finish", "label_stats", "gram",
                                                 "miner", "triplet_finalize", "sym_scale", "decode_loss", "cos_reduce",
                                                 "step_stats", "dh_gemm", "dh_finish", "dw_gemm", "bias_grads", "opt_step"};

struct dae_plan {
    bool prof;
    hipEvent_t ev0, ev1;
    // profile mode 2 (queued): one event pair per launch taken from this pool, the host never waits between launches; the pairs are read
    // when the pool cannot hold another step and by dae_plan_profile_read -- kernels and steps run back to back as they do un-profiled
    enum { PROF_POOL = 256, PROF_STEP_MAX = 48 };
    bool prof_queued;
    bool prof_stamps;                 // profile mode 3: the pairs carry the dispatches' own begin / end timestamps (DAE_LAUNCH, dae_common.h)
    int pev_used;
    hipEvent_t pev[PROF_POOL];
    int pev_slot[PROF_POOL / 2];
    // second stream for the miner chain (gram -> sweep -> finalize -> sym_scale): with batch_all the row weights
    // depend on the labels only, so the chain is independent of the decode GEMM and runs beside it
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    bool ev_dw_live;                  // ev_dw was recorded by the last dae_train_step (an event that never was recorded does not hold a waiter back)
    hipEvent_t ev_dw;                 // recorded right behind the kernel that completes the W gradient (dae_plan_dw_event): a data-parallel
                                      // caller starts its reduce-scatter from here, beside the step's tail kernel
    bool overlap_ok;
    int overlap_mode;                 // option "overlap" value: 1 = fork the decode before the Gram launch, 2 = after it (the Gram kernel needs a whole CU's LDS per workgroup
                                      // and cannot start beside resident decode workgroups), 3 = like 2 with the MINER on the side stream and the decode on the step's
    bool sym_ride_ok;                 // Gs = a/Nv (G + G^T) computed by rider workgroups of the decode launch instead of its own launch
    bool miner_order_ok;              // dispatch the batch_all workgroups by descending sweep cost (LabelJob::order)
    int32_t* miner_order;
    int32_t* cls_range;               // [1 + 2 Bpm]: sortedness flag + class range of every row (LabelJob::cls), the miner's range fast path
    bool miner_ranges_ok;             // option "miner_ranges" = 0: always compact positives / negatives by ballots
    double prof_ms[PS_COUNT];
    int prof_n[PS_COUNT];
    dae_config cfg;
    dae_buffers b;
    bool bound;
    int F, H, Fp, Hp, Bmax, Bpm;     // Bpm = padded max batch (leading dimension of every [.. x batch] image)
    int es;
    int s_enc, s_dh, s_gram;
    uint64_t ws_bytes;
    // carved pointers
    char *x, *xc, *xct, *h_lo, *h_t, *Gs, *delta2, *delta2_t, *delta1_t, *delta1_lo, *hcat_a, *hcat_b;
    // split-bf16 mode (dae_config.dtype = DAE_BF16X3): every stored operand x of the gradient GEMMs is hi + lo, both bf16; the *_2 images are the lo parts
    bool x3;
    char *W_lo2, *Wt_lo2, *h_t2, *delta2_2, *delta2_t2, *delta1_t2;
    char *x_2, *xct_2, *xc_2;         // ... and of the clean rows x / of x~^T / of x~ (dense input), used when the input values (or the corruption scale) are not exact in bf16
    int s_enc3, s_dh3;               // split-K slice counts of the dense-input encode / dh GEMMs in split-bf16 mode (3 resp. 4-5 K segments)
    // split-bf16 mode, the two lo product terms a CPU replay of the 20-step curve called droppable (tools/precision_study.py --per-term: cost 2.5e-5,
    // triplet 4.4e-5).  Measured on the GPU against the frozen reference curve, dropping them leaves the gate: cost 7.0e-5, triplet 1.56e-4
    // (profiles/r04_precision_terms.txt) -- so both stay ON; the options exist for that measurement (decode 57.9 -> 48.3 us without its term)
    // -> generalised to one bit per lo product term (X3T_* in dae_kernels.h, option "x3_terms"; the legacy options "x3_dec_wlo" / "x3_dh_hlo" flip their bit).
    // bf16 storage: all terms on.  fp16 storage (libdae_hip_f16.so): the two W terms alone (decode (h, W_lo), dh (delta2, W^T_lo)) -- no lo image of
    // delta2 / delta2^T / h / delta1 is written or read (CPU replay of the 20-step curve: cost 1.4e-5, triplet 6.5e-5; profiles/r04_precision_fp16_study.txt)
    uint32_t terms;
    // 16-bit images of the back-propagated operands (delta2, delta2^T, Gs, delta1^T) hold op_scale * value, a power of two the consuming epilogues
    // divide out (dh_finish: 1 / op_scale; the dW epilogue: OptEpi::gin): fp16's normal range ends at 6.1e-5 and delta2 ~ (y - x) / B, Gs ~ 1e-6 sit
    // below it.  1 for bf16 storage and fp32.  Option "op_scale_log2".
    float op_scale;
    int dec_bn;                      // tile width of the decode kernel: decode_tile_n(dtype), or 128 in the 16-bit modes when the 64-column tiles would be more than
                                     // DEC_WIDE_ROUNDS rounds of the chip's 768 slots (option "decode_bn" = 64 | 128 | 0 auto; before dae_plan_bind)
    bool dw_pair_ok;                 // option "dw_pair": split-bf16 dW kernel streams x~^T resp. delta2^T_hi ONCE for the hi and lo image of delta1^T resp. h^T
    bool xct2_clean;
    uint32_t* xtb;                   // x~^T as a bit image [Fp x Bpm/32] (binary CSR + bf16: operand of the sparse half of the dW kernel)
    bool xtb_clean;                  // the bit image holds only zeros (every step clears what it set; see step_tail_kernel)
    bool dw_bits_ok;               // option "dw_bits" = 0: dense x~^T image and a K = 2 Bp dW GEMM (A/B, equivalence tests)
    bool enc_w32_ok;                 // option "encode_w32": bf16 mode encodes from the fp32 MASTER weights (h fp32-accurate); 0 = from W_lo
    int w32_cols;                    // option "encode_w32_cols": 128 (default) or 64 H columns per workgroup of that kernel
    bool gram_split;                 // Gram matrix as a 3-term split-bf16 MFMA GEMM (bf16 mode) instead of exact-fp32 MFMA
    int dw_tr_mode;                  // option "dw_tr": the dW kernel reads x~ and delta2 ROW-MAJOR through transposing LDS reads (gemm_dw_pc<TRA>) -- the decode stores
                                     // delta2 once (no delta2^T), the gathers write x~ instead of x~^T.  1 on, 0 off, -1 (default) = on for DENSE train sets only:
                                     // measured (profiles/r05_ab_measurements.txt) -38 us per step at F = 50000 (the gather and the decode each write 90 MB less),
                                     // but +3..5 us at the CSR shape of c2 (11 fragment-read instructions per k step instead of 6; its decode does not get faster)
    bool gram64_ok;                  // option "gram64" (default 1): the split Gram on 64 x 64 tiles over the whole K, ONE slab (gram64_kernel); 0: 128 x 128 tiles, split-K
    float *slabs, *h_f32, *D_slabs, *G, *rowloss_part, *dbv_part, *colsum_part, *cos_part, *cos_stats, *cw, *loss_part,
        *dw_f32, *tri_scalars, *dh_extra, *rowsq_scratch, *tile_part, *zbuf;
    bool cos_zstore_ok;               // option "cos_zstore" (default 1): the cosine decode's second pass reads the first pass's accumulators back instead of recomputing the GEMM
    uint32_t *cnt_part, *role_cnt, *xc_bits, *x_bits;
    bool xbits_ok;                   // binary CSR + bf16: the decode epilogue reads x as a bit image (option "x_bits" = 0 disables)
    bool xct_clean;                  // x~^T holds only zeros (every step un-scatters what it wrote; see step_tail_kernel)
    bool tail_ok;                    // option "tail" = 0: separate bias_grads / step_stats launches and a full memset per step (A/B)
    bool fuse_opt_ok;                // option "fused_opt" = 0 keeps the separate optimizer kernel (A/B, equivalence tests)
    bool label_enc_ok;               // option "label_with_encode" = 0: label statistics ride on the gather launch / their own
    bool ce_literal;                 // option "ce_literal" = 1: cross_entropy always by the reference-literal formula
    bool sparse_ok;                  // CSR input: fused corrupt + gather + encode on the stored entries (option "encode_sparse" = 0: dense MFMA GEMM)
    bool bits_ok;                    // binary CSR + bf16: x~ handed to the encode GEMM as a bit image (dae_plan_set_option("encode_bits", 0) disables)
    int32_t *dw_i32, *n_same;
    int64_t *nvalid, *dw_i64;
    uint64_t* acc;
};

// lo image of the row-major shadow: exists (and is kept current by every kernel that updates W) only while the decode's (h, W_lo) term is on
static void* plan_w_lo2(const dae_plan* p) { return (p->x3 && (p->terms & X3T_DEC_WLO)) ? (void*)p->W_lo2 : nullptr; }

// tile width the decode launch of this plan uses (see dae_plan::dec_bn); lo images of delta2 / valued x keep the 64-column kernel
static int plan_dec_bn(const dae_plan* p) {
    const int def = decode_tile_n(p->cfg.dtype);
    if (p->es != 2) return def;
    const bool res = p->x3 && (p->terms & (X3T_DH_D2LO | X3T_DW_D2LO | X3T_XV));
    if (res) return def;
    if (p->dec_bn == 64 || p->dec_bn == 128) return p->dec_bn;
    const int64_t tiles64 = (int64_t)(p->Bpm / 128) * (p->Fp / 64);
    return tiles64 > 4 * 768 ? 128 : def;             // F = 50000: 5474 tiles of 128 x 64 = 7.1 rounds of 768 slots -> 2737 wide tiles
}

static int auto_splits(int tiles, int ktiles) {
    int s = 384 / (tiles > 0 ? tiles : 1);
    if (s >= 8) s = (s / 8) * 8;
    if (s > 16) s = 16;
    int cap = ktiles / 4;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return s;
}

// split-bf16 mode on dense-ndarray input: slice counts of the 3-segment encode and the 4-5-segment dh contraction (see dae_plan_create)
static void plan_x3_splits(dae_plan* p) {
    const dae_config& c = p->cfg;
    const int kt_f = p->Fp * p->es / 128, kt_b = p->Bpm * p->es / 128;
    p->s_enc3 = p->s_enc; p->s_dh3 = p->s_dh;
    if (!p->x3) return;
    const uint32_t T = p->terms;
    const int n_enc = 1 + ((T & X3T_ENC_WLO) ? 1 : 0) + ((T & X3T_ENC_XLO) ? 1 : 0);
    const int n_dh = 1 + ((T & X3T_DH_WLO) ? 1 : 0) + ((T & X3T_DH_D2LO) ? 1 : 0);
    if (c.encode_splits <= 0) if (const int w = gemm_w8_splits(c.dtype, p->Bpm, p->Hp, n_enc * kt_f)) p->s_enc3 = w;
    if (c.dh_splits <= 0) if (const int w = gemm_w8_splits(c.dtype, p->Bpm, p->Hp, n_dh * kt_f + ((T & X3T_DH_HLO) ? 2 : 1) * kt_b)) p->s_dh3 = w;
}

static uint64_t carve(dae_plan* p, char* base) {
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) -> char* {
        char* r = base ? base + off : nullptr;
        off += (bytes + 255) / 256 * 256;
        return r;
    };
    const uint64_t Bp = p->Bpm, Fp = p->Fp, Hp = p->Hp, es = p->es;
    p->x = take(Bp * Fp * es);
    p->xc = take(Bp * Fp * es);
    p->xct = take(Fp * Bp * es);
    p->xc_bits = (uint32_t*)take(Bp * (Fp / 32) * 4);
    p->x_bits = (uint32_t*)take(Bp * (Fp / 32) * 4);
    p->delta2 = take(Bp * Fp * es);
    p->delta2_t = take(Fp * Bp * es);
    int smax = p->s_enc > p->s_dh ? p->s_enc : p->s_dh;
    if (p->x3) { if (p->s_enc3 > smax) smax = p->s_enc3; if (p->s_dh3 > smax) smax = p->s_dh3; }
    p->slabs = (float*)take((uint64_t)smax * Bp * Hp * 4);
    p->h_f32 = (float*)take(Bp * Hp * 4);
    p->h_lo = take(Bp * Hp * es);
    p->h_t = take(Hp * Bp * es);
    p->delta1_t = take(Hp * Bp * es);
    p->delta1_lo = take(Bp * Hp * es);
    p->xtb = (uint32_t*)take(Fp * (Bp / 32) * 4);
    p->dh_extra = (float*)take(Bp * Hp * 4);
    const uint32_t T = p->x3 ? p->terms : 0u;          // lo images exist only for the product terms that read them
    p->W_lo2 = take((T & X3T_DEC_WLO) ? Fp * Hp * 2 : 256);
    p->Wt_lo2 = take(p->x3 ? Hp * Fp * 2 : 256);      // (the split mode's dW epilogue always writes it)
    p->h_t2 = take((T & (X3T_DH_HLO | X3T_DW_HLO)) ? Hp * Bp * 2 : 256);
    p->delta2_2 = take((T & X3T_DH_D2LO) ? Bp * Fp * 2 : 256);
    p->delta2_t2 = take((T & X3T_DW_D2LO) ? Fp * Bp * 2 : 256);
    p->delta1_t2 = take((T & X3T_DW_D1LO) ? Hp * Bp * 2 : 256);
    p->x_2 = take((T & X3T_XV) ? Bp * Fp * 2 : 256);
    p->xct_2 = take((T & X3T_XV) ? Fp * Bp * 2 : 256);
    p->xc_2 = take((T & X3T_ENC_XLO) ? Bp * Fp * 2 : 256);
    p->hcat_a = take(p->gram_split ? Bp * 3 * Hp * 2 : 256);
    p->hcat_b = take(p->gram_split ? Bp * 3 * Hp * 2 : 256);
    p->D_slabs = (float*)take((uint64_t)p->s_gram * Bp * Bp * 4);
    p->G = (float*)take(Bp * Bp * 4);
    p->Gs = take(Bp * Bp * es);
    p->role_cnt = (uint32_t*)take(Bp * Bp * 4);        // pos_triplets_only role counts (probe builds: the miner's timeline stamps)
    const uint64_t dbn = plan_dec_bn(p);                 // tile width of the decode kernel: lays out its partial-sum arrays
    p->rowloss_part = (float*)take((2 * Fp / dbn) * Bp * 4);
    p->dbv_part = (float*)take((2 * Bp / 128) * Fp * 4);
    p->colsum_part = (float*)take(2 * (Bp / 32) * Hp * 4);
    p->cos_part = (float*)take(2 * (2 * Fp / dbn) * Bp * 4);
    p->cos_stats = (float*)take(3 * Bp * 4);
    p->rowsq_scratch = (float*)take((Fp / 64) * Bp * 4);
    p->zbuf = (float*)take(p->cfg.loss_func == DAE_LOSS_COSINE ? Bp * Fp * 4 : 256);     // cosine: the decode's accumulators between its two passes (DecodeEpi::z_io)
    p->tile_part = (float*)take((Bp / 128) * (Fp / dbn) * 4);
    p->cw = (float*)take(Bp * 4);
    p->loss_part = (float*)take(Bp * 4);
    p->dw_f32 = (float*)take(Bp * 4);
    p->cnt_part = (uint32_t*)take(Bp * 4);
    p->dw_i32 = (int32_t*)take(Bp * 4);
    p->n_same = (int32_t*)take(Bp * 4);
    p->dw_i64 = (int64_t*)take(Bp * 8);
    p->nvalid = (int64_t*)take(256);
    p->acc = (uint64_t*)take(256);
    p->tri_scalars = (float*)take(256);
    p->miner_order = (int32_t*)take(Bp * 4);
    p->cls_range = (int32_t*)take((1 + 2 * Bp) * 4);
    return off;
}

extern "C" int dae_plan_create(const dae_config* cfg, dae_plan** out) {
    DAE_CHECK_ARG(cfg && out, "plan_create: null argument");
    DAE_CHECK_ARG(cfg->n_features > 0 && cfg->n_components > 0 && cfg->max_batch > 0, "plan_create: bad sizes");
    DAE_CHECK_ARG(cfg->dtype == DAE_BF16 || cfg->dtype == DAE_F32 || cfg->dtype == DAE_BF16X3, "plan_create: bad dtype");
    DAE_CHECK_ARG(cfg->enc_act >= 0 && cfg->enc_act <= 2 && cfg->dec_act >= 0 && cfg->dec_act <= 2, "plan_create: bad activation");
    DAE_CHECK_ARG(cfg->loss_func >= 0 && cfg->loss_func <= 2, "plan_create: bad loss_func");
    DAE_CHECK_ARG(cfg->opt >= 0 && cfg->opt <= 3, "plan_create: bad optimizer");
    DAE_CHECK_ARG(cfg->triplet >= 0 && cfg->triplet <= 3, "plan_create: bad triplet strategy");
    dae_plan* p = new (std::nothrow) dae_plan();
    DAE_CHECK_ARG(p, "plan_create: out of memory");
    memset(p, 0, sizeof(*p));
    p->cfg = *cfg;
    // split-bf16 mode: bf16 element type and kernels everywhere (cfg.dtype reads DAE_BF16 from here on); x3 adds the lo images, the
    // extra K segments of the three gradient GEMMs and the epilogues that write both parts
    p->x3 = cfg->dtype == DAE_BF16X3;
    if (p->x3) p->cfg.dtype = DAE_BF16;
    cfg = &p->cfg;
    p->F = cfg->n_features; p->H = cfg->n_components; p->Bmax = cfg->max_batch;
    p->Fp = (int)pad128(p->F); p->Hp = (int)pad128(p->H); p->Bpm = (int)pad128(p->Bmax);
    p->es = cfg->dtype == DAE_BF16 ? 2 : 4;
    const int tiles_bh = (p->Bpm / 128) * (p->Hp / 128);
    const int kt_f = p->Fp * p->es / 128;
    p->s_enc = cfg->encode_splits > 0 ? cfg->encode_splits : auto_splits(tiles_bh, kt_f);
    p->s_dh = cfg->dh_splits > 0 ? cfg->dh_splits : auto_splits(tiles_bh, kt_f);
    const int tiles_bb = (p->Bpm / 128) * (p->Bpm / 128);
    p->s_gram = cfg->gram_splits > 0 ? cfg->gram_splits : auto_splits(tiles_bb, p->Hp * 4 / 128);
    // large dense-input shapes: the 256 x 256 kernel picks its own slice count (one workgroup per CU)
    if (cfg->encode_splits <= 0) if (const int w = gemm_w8_splits(cfg->dtype, p->Bpm, p->Hp, kt_f)) p->s_enc = w;
    if (cfg->dh_splits <= 0) if (const int w = gemm_w8_splits(cfg->dtype, p->Bpm, p->Hp, kt_f + p->Bpm * p->es / 128)) p->s_dh = w;
    if (p->s_enc > kt_f) p->s_enc = kt_f;
    if (p->s_dh > kt_f) p->s_dh = kt_f;
    // split-bf16 mode on dense-ndarray input: the encode contraction has 3 K segments and dh 5; the 256 x 256 kernel is taken exactly when the
    // launch is handed ITS slice count for the real K-tile total, so these are planned with the segment lists' totals
    p->terms = kF16 ? (uint32_t)X3T_F16_DEFAULT : (uint32_t)X3T_ALL;
    // fp16 storage: op_scale = the largest power of two <= 16 * max_batch (capped at 2^14).  Bounds that keep the scaled images finite: |delta2| <= cw_i
    // (<= ~8 / B typically, <= 1 always) for a sigmoid decoder, |Gs| <= 2 alpha B / N_valid; the stores saturate at +-65504 beyond that (sat16)
    p->op_scale = 1.f;
    if (kF16 && p->es == 2) { float sc = 1.f; while (sc * 2.f <= 16.f * (float)p->Bmax && sc < 16384.f) sc *= 2.f; p->op_scale = sc; }
    p->dw_pair_ok = true;
    plan_x3_splits(p);
    if (p->s_gram > p->Hp * 4 / 128) p->s_gram = p->Hp * 4 / 128;
    p->gram_split = (cfg->dtype == DAE_BF16) && (p->x3 || cfg->triplet == DAE_TRIPLET_BATCH_ALL || cfg->triplet == DAE_TRIPLET_BATCH_HARD);   // x3: hcat_a also holds the row-major h_lo
    p->dw_tr_mode = -1;
    p->gram64_ok = p->gram_split && cfg->gram_splits <= 0;        // (an explicit split count keeps the 128 x 128 split-K form)
    if (p->gram64_ok) p->s_gram = 1;
    p->ws_bytes = carve(p, nullptr);
    // code-path choices below are plan state (dae_plan_set_option), never read from the environment
    p->fuse_opt_ok = true;
    p->tail_ok = true;
    p->label_enc_ok = true;
    p->ce_literal = false;
    p->xbits_ok = cfg->dtype == DAE_BF16;
    p->xct_clean = false; p->xtb_clean = false; p->xct2_clean = false;
    p->dw_bits_ok = false;                             // measured slower than streaming the dense image (profiles/r03_experiments.md)
    p->enc_w32_ok = cfg->dtype == DAE_BF16;
    p->w32_cols = 128;                                 // measured: 128-column fp32 slices (5.2 MB, served by the MALL) beat 64-column ones
    // binary CSR + bf16: x~ goes to the encode GEMM as a bit image whenever the 8-wave bit kernel can run the shape
    p->bits_ok = cfg->dtype == DAE_BF16 && encode_bits_fits(p->Bpm, p->Hp, p->Fp, p->s_enc);
    p->sparse_ok = true;
    p->miner_order_ok = true; p->sym_ride_ok = true; p->miner_ranges_ok = true;
    p->overlap_ok = false;                              // measured: running the miner chain beside decode is SLOWER (0.410 vs 0.351 ms/step)
    p->cos_zstore_ok = true;
    *out = p;
    return 0;
}

extern "C" void dae_plan_destroy(dae_plan* p) {
    if (!p) return;
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    for (int i = 0; i < dae_plan::PROF_POOL; ++i) if (p->pev[i]) (void)hipEventDestroy(p->pev[i]);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->ev_dw) (void)hipEventDestroy(p->ev_dw);
    if (p->side) (void)hipStreamDestroy(p->side);
    delete p;
}

// Code-path choices of a plan (A/B measurements and equivalence tests).  Every option selects between implementations of
// the SAME arithmetic; nothing here is read from the environment, so a stray variable can never change a training run.
extern "C" int dae_plan_set_option(dae_plan* p, const char* name, int32_t value) {
    DAE_CHECK_ARG(p && name, "plan_set_option: null argument");
    const bool on = value != 0;
    if (!strcmp(name, "encode_sparse")) p->sparse_ok = on;
    else if (!strcmp(name, "encode_bits")) p->bits_ok = on && p->cfg.dtype == DAE_BF16 && encode_bits_fits(p->Bpm, p->Hp, p->Fp, p->s_enc);
    else if (!strcmp(name, "x_bits")) p->xbits_ok = on && p->cfg.dtype == DAE_BF16;
    else if (!strcmp(name, "fused_opt")) p->fuse_opt_ok = on;
    else if (!strcmp(name, "dw_bits")) p->dw_bits_ok = on;
    else if (!strcmp(name, "dw_pair")) p->dw_pair_ok = on;
    else if (!strcmp(name, "dw_tr")) { DAE_CHECK_ARG(value >= -1 && value <= 1, "plan_set_option: dw_tr is -1 (auto), 0 or 1"); p->dw_tr_mode = value; p->xct_clean = false; }
    else if (!strcmp(name, "encode_w32")) p->enc_w32_ok = on && p->cfg.dtype == DAE_BF16;
    else if (!strcmp(name, "encode_w32_cols")) { DAE_CHECK_ARG(value == 64 || value == 128, "plan_set_option: encode_w32_cols is 64 or 128"); p->w32_cols = value; }
    else if (!strcmp(name, "tail")) p->tail_ok = on;
    else if (!strcmp(name, "label_with_encode")) p->label_enc_ok = on;
    else if (!strcmp(name, "ce_literal")) p->ce_literal = on;
    else if (!strcmp(name, "overlap")) { p->overlap_ok = on; p->overlap_mode = value; }
    else if (!strcmp(name, "gather_tile")) set_gather_tile(value);        // process-wide: tile shape of the dense gather (A/B measurements)
    else if (!strcmp(name, "dw_rounds")) { DAE_CHECK_ARG(value >= 1 && value <= 64, "plan_set_option: dw_rounds in 1..64"); set_use_glds(-100 - value); }   // process-wide, like miner_pack
    else if (!strcmp(name, "decode_pair")) set_use_glds(on ? -14 : -13);   // process-wide: the decode's two W terms as paired K-loop stages (one h tile, both W tiles)
    else if (!strcmp(name, "cos_zstore")) p->cos_zstore_ok = on;
    else if (!strcmp(name, "gram_fused")) set_use_glds(on ? -20 : -19);    // process-wide: the split Gram's three products per K tile in one LDS stage (gram64f_kernel; default on), 0 = the K-concatenated walk (gram64_kernel)
    else if (!strcmp(name, "decode_x3")) set_use_glds(on ? -18 : -17);     // process-wide: the split modes' decode on the K loops that keep the hi stage's fragments in registers (mainloop_n64_x3 / _c2; default on, binary input)
    else if (!strcmp(name, "decode_ast")) set_use_glds(on ? -16 : -15);    // process-wide: the A-stationary persistent decode kernel (gemm_decode_ast; default off: measured slower)
    else if (!strcmp(name, "pad_skip")) set_use_glds(on ? -12 : -11);      // process-wide: 0 = multiply / evaluate the all-padding 32-row blocks of the last batch tile too (A/B)
    else if (!strcmp(name, "miner_order")) p->miner_order_ok = on;
    else if (!strcmp(name, "miner_ranges")) p->miner_ranges_ok = on;
    else if (!strcmp(name, "miner_pack")) set_miner_pack(on);        // process-wide (the launcher's choice), like dae_set_glds
    else if (!strcmp(name, "miner_tile")) set_miner_tile(on);        // process-wide: 0 = the former wave-per-positive batch_all kernel
    else if (!strcmp(name, "sym_in_decode")) p->sym_ride_ok = on;
    else if (!strcmp(name, "x3_dec_wlo") || !strcmp(name, "x3_dh_hlo") || !strcmp(name, "x3_terms")) {
        DAE_CHECK_ARG(!p->bound, "plan_set_option: %s changes the split-K plan (workspace layout), set it before dae_plan_bind", name);
        if (name[3] == 't') { DAE_CHECK_ARG(value >= 0 && (uint32_t)value <= X3T_ALL, "plan_set_option: x3_terms is a mask of the X3T_* bits (0..%u)", (unsigned)X3T_ALL); p->terms = (uint32_t)value; }
        else { const uint32_t bit = name[4] == 'e' ? X3T_DEC_WLO : X3T_DH_HLO; p->terms = on ? (p->terms | bit) : (p->terms & ~bit); }
        plan_x3_splits(p);
        p->ws_bytes = carve(p, nullptr);
    }
    else if (!strcmp(name, "decode_bn")) {
        DAE_CHECK_ARG(!p->bound, "plan_set_option: decode_bn changes the workspace layout, set it before dae_plan_bind");
        DAE_CHECK_ARG(value == 0 || value == 64 || value == 128, "plan_set_option: decode_bn is 0 (auto), 64 or 128");
        p->dec_bn = value;
        p->ws_bytes = carve(p, nullptr);
    }
    else if (!strcmp(name, "op_scale_log2")) {
        DAE_CHECK_ARG(value >= 0 && value <= 20, "plan_set_option: op_scale_log2 in 0..20");
        DAE_CHECK_ARG(p->es == 2, "plan_set_option: op_scale_log2 applies to the 16-bit modes");
        p->op_scale = (float)(1u << value);
    }
    else if (!strcmp(name, "gram64")) {
        DAE_CHECK_ARG(!p->bound, "plan_set_option: gram64 changes the workspace layout (slab count), set it before dae_plan_bind");
        p->gram64_ok = on && p->gram_split;
        const int tiles_bb = (p->Bpm / 128) * (p->Bpm / 128);
        p->s_gram = p->gram64_ok ? 1 : (p->cfg.gram_splits > 0 ? p->cfg.gram_splits : auto_splits(tiles_bb, p->Hp * 4 / 128));
        if (p->s_gram > p->Hp * 4 / 128) p->s_gram = p->Hp * 4 / 128;
        p->ws_bytes = carve(p, nullptr);
    }
    else if (!strcmp(name, "gram_fp32")) {
        DAE_CHECK_ARG(!p->bound, "plan_set_option: gram_fp32 changes the workspace layout, set it before dae_plan_bind");
        DAE_CHECK_ARG(!p->x3, "plan_set_option: gram_fp32 is not available in split-bf16 mode (its Gram operands double as the row-major h images)");
        p->gram_split = !on && p->cfg.dtype == DAE_BF16 && (p->cfg.triplet == DAE_TRIPLET_BATCH_ALL || p->cfg.triplet == DAE_TRIPLET_BATCH_HARD);
        if (!p->gram_split && p->gram64_ok) {          // the exact-fp32 Gram runs on the 128 x 128 split-K kernel: its slab count again
            p->gram64_ok = false;
            const int tiles_bb = (p->Bpm / 128) * (p->Bpm / 128);
            p->s_gram = p->cfg.gram_splits > 0 ? p->cfg.gram_splits : auto_splits(tiles_bb, p->Hp * 4 / 128);
            if (p->s_gram > p->Hp * 4 / 128) p->s_gram = p->Hp * 4 / 128;
        }
        p->ws_bytes = carve(p, nullptr);
    } else {
        set_error("plan_set_option: unknown option '%s'", name);
        return 1;
    }
    return 0;
}

static int prof_flush(dae_plan* p);
extern "C" int dae_plan_profile(dae_plan* p, int32_t enable) {
    DAE_CHECK_ARG(p, "plan_profile: null plan");
    if (enable && !p->ev0) {
        DAE_CHECK_HIP(hipEventCreate(&p->ev0));
        DAE_CHECK_HIP(hipEventCreate(&p->ev1));
    }
    const int rf = p->pev_used ? prof_flush(p) : 0;     // pairs still queued belong to the mode being left; a failed read is reported, the switch still happens
    if ((enable == 2 || enable == 3) && !p->pev[0])
        for (int i = 0; i < dae_plan::PROF_POOL; ++i) DAE_CHECK_HIP(hipEventCreate(&p->pev[i]));
    if (enable) { memset(p->prof_ms, 0, sizeof(p->prof_ms)); memset(p->prof_n, 0, sizeof(p->prof_n)); }
    p->prof = enable != 0;
    p->prof_queued = enable == 2 || enable == 3;
    p->prof_stamps = enable == 3;
    p->pev_used = 0;
    return rf;
}

extern "C" int dae_plan_profile_read(const dae_plan* p, int32_t max_slots, double* ms_total, int32_t* counts) {
    DAE_CHECK_ARG(p && ms_total && counts, "plan_profile_read: null argument");
    if (int rf = prof_flush(const_cast<dae_plan*>(p))) return rf;
    for (int i = 0; i < PS_COUNT && i < max_slots; ++i) { ms_total[i] = p->prof_ms[i]; counts[i] = p->prof_n[i]; }
    return PS_COUNT <= max_slots ? 0 : 1;
}

extern "C" int32_t dae_plan_profile_slots(void) { return PS_COUNT; }
extern "C" const char* dae_plan_profile_name(int32_t slot) { return (slot >= 0 && slot < PS_COUNT) ? kProfNames[slot] : ""; }
extern "C" uint64_t dae_plan_workspace_bytes(const dae_plan* p) { return p ? p->ws_bytes : 0; }

extern "C" int dae_plan_bind(dae_plan* p, const dae_buffers* bufs) {
    DAE_CHECK_ARG(p && bufs, "plan_bind: null argument");
    DAE_CHECK_ARG(bufs->W && bufs->bh && bufs->bv && bufs->grad && bufs->W_lo && bufs->Wt_lo, "plan_bind: null parameter buffer");
    DAE_CHECK_ARG(bufs->workspace && bufs->workspace_bytes >= p->ws_bytes, "plan_bind: workspace too small (%llu < %llu)",
                  (unsigned long long)bufs->workspace_bytes, (unsigned long long)p->ws_bytes);
    DAE_CHECK_ARG(((uintptr_t)bufs->workspace % 256) == 0, "plan_bind: workspace must be 256-byte aligned");
    DAE_CHECK_ARG((bufs->indptr != nullptr) != (bufs->dense != nullptr) || (!bufs->indptr && !bufs->dense),
                  "plan_bind: give either a CSR or a dense train set");
    DAE_CHECK_ARG(p->cfg.opt == DAE_OPT_SGD || bufs->opt_s1, "plan_bind: optimizer slot buffer required");
    DAE_CHECK_ARG(p->cfg.opt != DAE_OPT_ADAM || bufs->opt_s2, "plan_bind: second optimizer slot buffer required");
    p->b = *bufs;
    carve(p, (char*)bufs->workspace);
    p->bound = true;
    p->xct_clean = false;            // a (re)bound workspace has not been cleared: the next backward step memsets x~^T once
    p->xtb_clean = false; p->xct2_clean = false;
    return 0;
}

extern "C" int dae_plan_sync_shadows(dae_plan* p, void* stream) {
    DAE_CHECK_ARG(p && p->bound, "plan_sync_shadows: plan not bound");
    return launch_opt_step(p->cfg.opt, 0.f, 0.f, 1.f, p->b.W, p->b.bh, p->b.bv, p->b.grad, p->b.opt_s1, p->b.opt_s2, p->Fp, p->Hp,
                           p->cfg.dtype, p->b.W_lo, p->b.Wt_lo, plan_w_lo2(p), p->x3 ? p->Wt_lo2 : nullptr, /*apply=*/0, stream);
}

extern "C" void* dae_plan_buffer(dae_plan* p, const char* name) {
    if (!p || !p->bound || !name) return nullptr;
#define DAE_BUF(n) if (!strcmp(name, #n)) return (void*)p->n;
    DAE_BUF(hcat_a) DAE_BUF(hcat_b) DAE_BUF(x) DAE_BUF(xc) DAE_BUF(xct) DAE_BUF(h_lo) DAE_BUF(h_t) DAE_BUF(Gs) DAE_BUF(delta2) DAE_BUF(delta2_t) DAE_BUF(delta1_t)
    DAE_BUF(delta1_lo) DAE_BUF(xtb) DAE_BUF(W_lo2) DAE_BUF(Wt_lo2) DAE_BUF(h_t2) DAE_BUF(delta2_2) DAE_BUF(delta2_t2) DAE_BUF(delta1_t2) DAE_BUF(x_2) DAE_BUF(xct_2) DAE_BUF(xc_2)
    DAE_BUF(slabs) DAE_BUF(h_f32) DAE_BUF(D_slabs) DAE_BUF(G) DAE_BUF(rowloss_part) DAE_BUF(dbv_part) DAE_BUF(colsum_part)
    DAE_BUF(cos_part) DAE_BUF(cos_stats) DAE_BUF(cw) DAE_BUF(loss_part) DAE_BUF(dw_f32) DAE_BUF(tri_scalars) DAE_BUF(dh_extra)
    DAE_BUF(tile_part) DAE_BUF(cnt_part) DAE_BUF(role_cnt) DAE_BUF(dw_i32) DAE_BUF(n_same) DAE_BUF(nvalid) DAE_BUF(dw_i64)
#undef DAE_BUF
    return nullptr;
}

extern "C" int dae_plan_info(const dae_plan* p, int32_t* out8) {
    DAE_CHECK_ARG(p && out8, "plan_info: null");
    out8[0] = p->Fp; out8[1] = p->Hp; out8[2] = p->Bpm; out8[3] = p->s_enc; out8[4] = p->s_dh; out8[5] = p->s_gram;
    out8[6] = p->es; out8[7] = p->x3 ? (int32_t)(1u | (p->terms << 1) | ((uint32_t)ilogbf(p->op_scale) << 16)) : (int32_t)((uint32_t)ilogbf(p->op_scale) << 16);   // bit 0: split mode; bits 1..11: its lo terms; bits 16..: log2 op_scale
    return 0;
}

static int gather_batch(dae_plan* p, const int64_t* indptr, const int32_t* indices, const float* values, const float* dense,
                        int64_t ld_dense, const int32_t* row_idx, int B, void* x, void* xc, void* xct, float* rowsq,
                        int corr_mode, const uint32_t* keep_bits, uint64_t seed, uint32_t rng_stream, float corr_frac,
                        float scale, void* stream, uint32_t* xc_bits = nullptr, const LabelJob* label_job = nullptr,
                        uint32_t* x_bits = nullptr, void* x2 = nullptr) {
    if (indptr)
        return launch_gather_csr(indptr, indices, values, row_idx, B, p->F, p->cfg.dtype, x, xc, p->Fp, xct, p->Bpm, rowsq, corr_mode,
                                 keep_bits, seed, rng_stream, corr_frac, scale, xc_bits, p->Fp / 32, label_job, (hipStream_t)stream, x_bits, x2);
    DAE_CHECK_ARG(dense, "step: no train set bound");
    return dae_gather_dense(dense, ld_dense, row_idx, B, p->F, p->cfg.dtype, x, xc, p->Fp, xct, p->Bpm, rowsq, p->rowsq_scratch,
                            corr_mode, keep_bits, seed, rng_stream, corr_frac, scale, stream);
}
// split-bf16 mode, dense input: the lo images of x / x~ / x~^T (a second pass over the fp32 rows with the same keep decisions)
static int gather_dense_lo(dae_plan* p, const float* dense, int64_t ld_dense, const int32_t* row_idx, int B, void* x2, void* xc2, void* xct2,
                           int corr_mode, const uint32_t* keep_bits, uint64_t seed, uint32_t rng_stream, float corr_frac, float scale, void* stream) {
    return launch_gather_dense(dense, ld_dense, row_idx, B, p->F, p->cfg.dtype, x2, xc2, p->Fp, xct2, p->Bpm, nullptr, nullptr, corr_mode, keep_bits,
                               seed, rng_stream, corr_frac, scale, stream, 1);
}

#define RC(expr) do { if (int rc__ = (expr)) return rc__; } while (0)
// K5: D = h h^T (triplet_loss_utils.py:93,219).  fp32 mode: exact-fp32 MFMA.  bf16 mode: split-bf16 (h = hi + lo,
// three bf16 MFMA products concatenated along K = 3*Hp), ~2^-17 relative error, 16x the MFMA rate.
// learning rate handed to the optimizer kernels (Adam: lr_t = lr * sqrt(1-b2^t)/(1-b1^t), TF AdamOptimizer)
static float plan_lr(const dae_plan* p, int adam_t) {
    float lr = p->cfg.learning_rate;
    if (p->cfg.opt == DAE_OPT_ADAM) {
        const double t = adam_t < 1 ? 1 : adam_t;
        lr = (float)((double)lr * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
    }
    return lr;
}
static int launch_gram(dae_plan* p, int Bp, int Hp, int64_t dslab, hipStream_t st) {
    if (p->gram_split && p->gram64_ok && p->s_gram == 1) return launch_gram64(p->hcat_a, p->hcat_b, Bp, Hp, p->D_slabs, st);
    if (p->gram_split)
        return launch_gemm_f32out(DAE_BF16, Bp, Bp, p->hcat_a, 3 * Hp, p->hcat_b, 3 * Hp, 3 * Hp, nullptr, 0, nullptr, 0, 0, p->D_slabs, Bp,
                                  p->s_gram, dslab, st, GEMM_ROLE_GRAM);
    return launch_gemm_f32out(DAE_F32, Bp, Bp, p->h_f32, Hp, p->h_f32, Hp, Hp, nullptr, 0, nullptr, 0, 0, p->D_slabs, Bp, p->s_gram, dslab, st,
                              GEMM_ROLE_GRAM);
}
// PROF(slot, call): in profile mode time the call with HIP events ON THE STEP'S STREAM and accumulate the elapsed GPU time of that slot
// (dae_plan_profile, include/dae_hip.h).  Mode 1: an event pair around the call and a host wait behind it.  Mode 2 (q__): a pair of the
// plan's pool around the call, read later by prof_flush.  Mode 3 (k__): the pool is handed to DAE_LAUNCH (dae_common.h) for the duration
// of the call, every kernel launch inside takes a pair and has it stamped by its own dispatch; memsets keep the mode-2 form.
#define PROF(slot, expr)                                                            \
    do {                                                                            \
        const bool q__ = p->prof && p->prof_queued && p->pev_used + 2 <= dae_plan::PROF_POOL; \
        const bool k__ = q__ && p->prof_stamps && (slot) != PS_MEMSET;              \
        if (k__) g_lt = LaunchTimer{p->pev, &p->pev_used, p->pev_slot, dae_plan::PROF_POOL, (slot), true}; \
        else if (q__) DAE_CHECK_HIP(hipEventRecord(p->pev[p->pev_used], st));       \
        else if (p->prof) DAE_CHECK_HIP(hipEventRecord(p->ev0, st));                \
        const int rc_prof__ = (expr);                                               \
        if (k__) g_lt.pool = nullptr;                                               \
        if (rc_prof__) return rc_prof__;                                            \
        if (k__) break;                     /* the launches took their pairs */     \
        if (q__) {                                                                  \
            DAE_CHECK_HIP(hipEventRecord(p->pev[p->pev_used + 1], st));             \
            p->pev_slot[p->pev_used / 2] = (slot) | 0x100; p->pev_used += 2;        \
        } else if (p->prof) {                                                       \
            DAE_CHECK_HIP(hipEventRecord(p->ev1, st));                              \
            DAE_CHECK_HIP(hipEventSynchronize(p->ev1));                             \
            float ms__ = 0.f;                                                       \
            DAE_CHECK_HIP(hipEventElapsedTime(&ms__, p->ev0, p->ev1));              \
            p->prof_ms[slot] += ms__; p->prof_n[slot] += 1;                         \
        }                                                                           \
    } while (0)
// queued profile mode: wait for the step's last pair, then add every pair to its slot
static int prof_flush(dae_plan* p) {
    if (!p->prof_queued || p->pev_used == 0) return 0;
    const int used = p->pev_used;
    p->pev_used = 0;                                    // (also on the error paths below: a failed read must not poison the next profile call)
    DAE_CHECK_HIP(hipEventSynchronize(p->pev[used - 1]));
    for (int i = 0; i < used; i += 2) {
        float ms = 0.f;
        DAE_CHECK_HIP(hipEventElapsedTime(&ms, p->pev[i], p->pev[i + 1]));
        const int sl = p->pev_slot[i / 2] & 0xff;          // bit 8: the first launch of its PROF call (a call with several launches counts once)
        p->prof_ms[sl] += ms; p->prof_n[sl] += (p->pev_slot[i / 2] >> 8) & 1;
    }
    return 0;
}
static int memset_async(void* ptr, size_t bytes, hipStream_t st) {
    DAE_CHECK_HIP(hipMemsetAsync(ptr, 0, bytes, st));
    return 0;
}

static int train_step_body(dae_plan* p, const dae_step* s, void* stream);
extern "C" int dae_train_step(dae_plan* p, const dae_step* s, void* stream) {
    const int rc = train_step_body(p, s, stream);
    if (p && p->prof_queued && p->pev_used + dae_plan::PROF_STEP_MAX > dae_plan::PROF_POOL) { const int rf = prof_flush(p); return rc ? rc : rf; }
    return rc;
}
static int train_step_body(dae_plan* p, const dae_step* s, void* stream) {
    DAE_CHECK_ARG(p && p->bound && s, "train_step: plan not bound / null step");
    DAE_CHECK_ARG(s->row_idx && s->B > 0 && s->B <= p->Bmax, "train_step: batch %d outside (0, %d]", s ? s->B : -1, p->Bmax);
    const dae_config& c = p->cfg;
    const bool explicit3 = (c.triplet == 3);
    DAE_CHECK_ARG(c.triplet == DAE_TRIPLET_NONE || explicit3 || s->labels, "train_step: labels required for triplet mining");
    DAE_CHECK_ARG(!explicit3 || s->B % 3 == 0, "train_step: explicit-triplet batch must stack org/pos/neg (B %% 3 == 0)");
    DAE_CHECK_ARG(s->stats, "train_step: stats pointer required");
    hipStream_t st = (hipStream_t)stream;
    const int B = s->B, Bp = (int)pad128(B), F = p->F, H = p->H, Fp = p->Fp, Hp = p->Hp, ldB = p->Bpm, dt = c.dtype;
    const bool is_cos = c.loss_func == DAE_LOSS_COSINE;
    const bool backward = s->phase != 2;
    // phases 4 / 5 split the step around an EXTERNAL miner (data parallel with global-batch mining, dp.GlobalMiner): phase 4
    // stops after the encode (h_f32 / h_lo / h_t and the side images stay in the workspace); the caller mines over the
    // all-gathered batch and writes the row weights (cw), the triplet scalars and d(triplet)/dh (dh_extra) into the plan's
    // buffers; phase 5 resumes at the decode and ends like phase 1 (gradients in the flat buffer).
    const bool h_only = s->phase == 4, resume = s->phase == 5, ext_mine = h_only || resume;

    // 1-2. corrupt + gather  (K0/K1 front half)
    // x~^T: the CSR gather only scatters kept entries, so the image must be zero beforehand.  The step tail un-scatters
    // exactly what was written, so the 18 MB memset runs once (or after a failed / foreign step); the dense gather
    // overwrites whole tiles and never needs it.
    const bool csr_in = s->c_indptr || p->b.indptr;
    // label statistics (cw, N_valid, data weights) depend on the labels alone: they ride on the CSR gather launch
    LabelJob lj{s->labels, B, Bp, c.triplet, p->nvalid, p->dw_i64, p->cw, c.alpha, p->tri_scalars, p->miner_order_ok ? p->miner_order : nullptr,
                p->miner_ranges_ok ? p->cls_range : nullptr};
    // ... on the encode GEMM's launch when that grid leaves a CU free (else on the CSR gather's, else their own)
    const bool label_with_encode = p->tail_ok && !explicit3 && !ext_mine && Bp <= 1024 && p->label_enc_ok;
    const bool label_in_gather = p->tail_ok && !label_with_encode && !explicit3 && !ext_mine && !s->c_indptr && p->b.indptr && Bp <= 1024;
    const bool tail = p->tail_ok;
    float* rowsq = is_cos ? p->cos_stats : nullptr;
    bool use_bits = false;
    // binary CSR train set in bf16 mode: the clean rows reach the decode epilogue as a bit image (1.1 MB, not 18 MB)
    const bool use_xbits = p->xbits_ok && p->b.indptr && !p->b.values;
    const bool use_sparse = p->sparse_ok && csr_in;
    const bool dense_in = !csr_in && p->b.dense;
    // phase 0 / 3 in bf16 mode: the optimizer runs in the dW GEMM's epilogue (phase 3 does not materialise the W gradient);
    // phase 1 / 5 (data parallel) in bf16 mode: the same kernel in its gradient-only form when the shape fits it
    const bool apply_now = (s->phase == 0 || s->phase == 3);
    const bool fuse_opt0 = backward && apply_now && dt == DAE_BF16 && p->fuse_opt_ok;
    // binary CSR + bf16 + the fused sparse encode: x~^T is a BIT image (1.1 MB) from which the dW kernel's producer waves build the
    // A tiles of its x~^T.delta1 segment in LDS; otherwise the dense x~^T image (18 MB, scattered / un-scattered every step) is streamed
    const bool src_binary = s->c_indptr ? !s->c_values : (p->b.indptr && !p->b.values);
    const bool x3 = p->x3;
    const uint32_t T = x3 ? p->terms : 0u;                   // lo product terms that are multiplied (X3T_*)
    const float osc = p->es == 2 ? p->op_scale : 1.f, oinv = 1.f / osc;   // operand scale of the 16-bit delta images (a power of two)
    // split-bf16 mode: the fused dW + optimizer kernel exists for shapes of at most one 160 x 128 tile per CU; larger shapes (and the
    // data-parallel gradient-only phases) take the N-segment dW GEMM to memory + the optimizer kernel that writes all four shadows
    const bool fuse_opt = fuse_opt0 && (!x3 || dw_x3_fits(Fp, Hp, Bp));
    // (split-bf16 mode: the gradient-only form of the same N-segment kernel, fp32 gradient to the flat buffer)
    const bool dw_pc_grad = backward && !apply_now && dt == DAE_BF16 && p->fuse_opt_ok && (x3 ? dw_x3_fits(Fp, Hp, Bp) : dw_bits_fits(Fp, Hp, Bp));
    const bool dw_bits = p->dw_bits_ok && use_sparse && src_binary && backward && dt == DAE_BF16 && (fuse_opt || dw_pc_grad) &&
                           dw_bits_fits(Fp, Hp, Bp);
    // contractions over the BATCH (dW's K, the Gs.h segment of dh) stop at the last 64-deep K tile that holds a real row: the images are zero beyond B, and
    // B = 800 pads to 896 = 14 K tiles of which 13 hold data
    const int Bk = (B + 63) / 64 * 64;
    // Transposed-A dW (gemm_dw_pc<TRA>): x~ and delta2 are consumed ROW-MAJOR [batch x feature], so delta2^T is never stored and the gathers write x~ (the
    // CSR scatter lands in one 20 KB row per batch row instead of one line per entry).  Needs the 160 x 128 kernel and the two plain K segments
    // (16-bit modes without lo images of x~ / delta2 / delta1 / h in dW: f16x2, bf16, f16); decided from plan state and shapes only, so that the
    // phase-4 / phase-5 halves of an externally mined step agree
    const bool dw_plain2 = !x3 || !(T & (X3T_DW_D1LO | X3T_DW_HLO | X3T_DW_D2LO | X3T_XV));
    const bool dw_tr = (p->dw_tr_mode == 1 || (p->dw_tr_mode < 0 && dense_in)) && dt == DAE_BF16 && backward && (fuse_opt || dw_pc_grad) && dw_plain2 && !dw_bits && (use_sparse || dense_in) &&
                       (x3 || dw_pc_taken(Fp, Hp, Bk, Bk, !fuse_opt));
    if (x3) {
        // split-bf16 mode: CSR input encoded from the fp32 master weights (h is fp32-accurate and its hi / lo images come from the same
        // launch); x~ must be exact in bf16 (binary data, or values with <= 8 significant bits).  Every phase: the data-parallel
        // exchange of this mode moves fp32 gradients and fp32 master rows (dp.ShardedExchange), so the master is current on every rank
        // (h from the 16-bit hi image of W alone was measured in round 6, profiles/r06_ab_measurements.txt: the same 26.8 us -- the kernel is not bound by its
        //  W-row bytes -- and the triplet leg of c2 leaves the gate at step 5 (1.5e-3): refused, not offered)
        DAE_CHECK_ARG((use_sparse && p->enc_w32_ok) || dense_in, "train_step: split-bf16 mode needs the fp32-master sparse encode (CSR input) or a dense train set");
        DAE_CHECK_ARG(!p->b.grad_lo, "train_step: split-bf16 mode exchanges fp32 gradients (no bf16 gradient image)");
        DAE_CHECK_ARG(!dw_bits, "train_step: split-bf16 mode streams the dense x~^T image (option dw_bits off)");
    }
    // split-bf16 mode with VALUED input (tf-idf, salt-and-pepper copies, decay noise's scale factor): x~ = scale * v is not exact in bf16, so
    // x~^T and the clean rows x get lo images too (xct_2, x_2) and the dW contraction walks 6 segments; binary data with a bf16-exact scale
    // (masking noise: 1.0) needs neither
    bool x3_vals = false;
    if (x3) {
        uint32_t u; memcpy(&u, &s->scale, 4);
        x3_vals = (T & X3T_XV) && (!src_binary || (u & 0xffffu) != 0u || (p->b.indptr && p->b.values) || dense_in);
    }
    const bool x2_clean = x3 && (T & X3T_XV) && !use_xbits && (p->b.values || p->b.dense);    // the clean rows get a lo image (valued CSR / dense train set)
    if (!resume && backward && csr_in) {
        if (dw_bits) { if (!(tail && p->xtb_clean)) PROF(PS_MEMSET, memset_async(p->xtb, (size_t)Fp * (ldB / 32) * 4, st)); }
        else if (!(tail && p->xct_clean)) PROF(PS_MEMSET, memset_async(p->xct, (size_t)Fp * ldB * p->es, st));
        if (x3_vals && !(tail && p->xct2_clean)) PROF(PS_MEMSET, memset_async(p->xct_2, (size_t)Fp * ldB * 2, st));
    }
    if (backward) { if (dw_bits) p->xtb_clean = false; else p->xct_clean = false; if (x3_vals) p->xct2_clean = false; }
    const int64_t slab = (int64_t)Bp * Hp;
    int enc_label_done = 0;
    bool labels_done = ext_mine;               // label statistics already produced by a workgroup of an earlier launch (or by the caller)
    if (resume) {
        // h and the side images are those of the preceding phase-4 call
    } else if (use_sparse) {
        // CSR input: corrupt + gather + encode in one launch on the stored entries (tf.sparse.matmul, autoencoder.py:377,389);
        // the dense x~ image is never formed.  The clean rows reach the decode epilogue as a bit image written by the same
        // launch (binary data) or as a dense tile from the gather kernel (valued data / explicitly corrupted copy).
        const bool w32 = p->enc_w32_ok && dt == DAE_BF16;   // (a sharded-optimizer exchange turns the option off: only W_lo is current on every rank)
        // the encode launch also emits the clean-row images, unless their LDS rows do not fit (e.g. 50000 features)
        const bool own_clean = !s->c_indptr && use_xbits && encode_csr_lds_bytes(dt, w32, p->w32_cols, Fp / 32) <= 64 * 1024;
        if (!own_clean)
            PROF(PS_GATHER, gather_batch(p, p->b.indptr, p->b.indices, p->b.values, p->b.dense, p->b.ld_dense, s->row_idx, B,
                            use_xbits ? nullptr : p->x, nullptr, nullptr, rowsq, DAE_CORR_NONE, nullptr, 0, 0, 0.f, 1.f, stream, nullptr, nullptr,
                            use_xbits ? p->x_bits : nullptr, (x2_clean && p->b.values) ? p->x_2 : nullptr));
        // a DENSE train set with an explicitly corrupted CSR copy (salt-and-pepper): the clean rows' lo image comes from the dense rows
        if (!own_clean && x2_clean && p->b.dense && !p->b.indptr)
            PROF(PS_GATHER, gather_dense_lo(p, p->b.dense, p->b.ld_dense, s->row_idx, B, p->x_2, nullptr, nullptr, DAE_CORR_NONE, nullptr, 0, 0, 0.f, 1.f, stream));
        EncCsrLaunch q;
        memset(&q, 0, sizeof(q));
        q.indptr = s->c_indptr ? s->c_indptr : p->b.indptr; q.indices = s->c_indptr ? s->c_indices : p->b.indices;
        q.values = s->c_indptr ? s->c_values : p->b.values; q.row_idx = (s->c_indptr && s->c_row_idx) ? s->c_row_idx : s->row_idx; q.B = B; q.F = F; q.H = H; q.dtype = dt;
        q.W = w32 ? (const void*)p->b.W : (const void*)p->b.W_lo; q.w_f32 = w32 ? 1 : 0; q.w32_cols = p->w32_cols; q.ldw = Hp; q.bh = p->b.bh; q.enc_act = c.enc_act;
        q.corr_mode = s->c_indptr ? DAE_CORR_NONE : s->corr_mode; q.keep_bits = s->keep_bits; q.seed = s->seed; q.rng_stream = s->rng_stream;
        q.corr_frac = s->corr_frac; q.scale = s->scale;
        q.h_f32 = p->h_f32; q.h_lo = p->h_lo; q.ldh = Hp; q.h_t = p->h_t; q.ldht = ldB;
        q.hcat_a = p->gram_split ? p->hcat_a : nullptr; q.hcat_b = p->gram_split ? p->hcat_b : nullptr;
        q.x_bits = own_clean ? p->x_bits : nullptr; q.ldxb = Fp / 32; q.xct = (backward && !dw_bits) ? p->xct : nullptr; q.ldt = dw_tr ? Fp : ldB;
        q.xct_rm = dw_tr ? 1 : 0;
        q.xtb = (backward && dw_bits) ? p->xtb : nullptr; q.ldxt = ldB / 32;
        q.rowsq = own_clean ? rowsq : nullptr;
        q.h_t2 = (T & (X3T_DH_HLO | X3T_DW_HLO)) ? p->h_t2 : nullptr;
        q.xct2 = (x3_vals && backward) ? p->xct_2 : nullptr;
        q.label_job = label_with_encode ? &lj : nullptr;
        PROF(PS_ENC_GEMM, launch_encode_csr(q, st));
        labels_done = label_with_encode;
    } else {
        if (s->c_indptr) {   // an explicitly corrupted copy of the train set (salt&pepper, host-side noise)
            PROF(PS_GATHER, gather_batch(p, p->b.indptr, p->b.indices, p->b.values, p->b.dense, p->b.ld_dense, s->row_idx, B, use_xbits ? nullptr : p->x,
                            nullptr, nullptr, rowsq, DAE_CORR_NONE, nullptr, 0, 0, 0.f, 1.f, stream, nullptr, nullptr, use_xbits ? p->x_bits : nullptr));
            PROF(PS_GATHER, gather_batch(p, s->c_indptr, s->c_indices, s->c_values, nullptr, 0, s->c_row_idx ? s->c_row_idx : s->row_idx, B, nullptr, p->xc,
                            backward ? p->xct : nullptr, nullptr, DAE_CORR_NONE, nullptr, 0, 0, 0.f, s->scale, stream));
        } else {
            // binary CSR, unit scale, bf16: the corrupted batch goes to the encode GEMM as a BIT image (1.1 MB, not 18 MB of bf16)
            use_bits = p->bits_ok && p->b.indptr && !p->b.values && s->scale == 1.0f;
            PROF(PS_GATHER, gather_batch(p, p->b.indptr, p->b.indices, p->b.values, p->b.dense, p->b.ld_dense, s->row_idx, B,
                            use_xbits ? nullptr : p->x, use_bits ? nullptr : p->xc, (backward && !dw_tr) ? p->xct : nullptr, rowsq, s->corr_mode, s->keep_bits,
                            s->seed, s->rng_stream, s->corr_frac, s->scale, stream, use_bits ? p->xc_bits : nullptr, label_in_gather ? &lj : nullptr,
                            use_xbits ? p->x_bits : nullptr));
        }
        if (x3 && dense_in && (T & (X3T_XV | X3T_ENC_XLO)))
            PROF(PS_GATHER, gather_dense_lo(p, p->b.dense, p->b.ld_dense, s->row_idx, B, (T & X3T_XV) ? p->x_2 : nullptr, (T & X3T_ENC_XLO) ? p->xc_2 : nullptr,
                                            (backward && (T & X3T_XV)) ? p->xct_2 : nullptr, s->corr_mode,
                                            s->keep_bits, s->seed, s->rng_stream, s->corr_frac, s->scale, stream));
        // 3-4. encode (K1/K2)
        if (x3 && dense_in) {     // z1 = x~ W as (x~_hi, W^T_hi) (x~_hi, W^T_lo) (x~_lo, W^T_hi)
            const GemmSegDesc es3[3] = {{p->xc, Fp, p->b.Wt_lo, Fp, Fp}, {p->xc, Fp, p->Wt_lo2, Fp, (T & X3T_ENC_WLO) ? Fp : 0},
                                        {p->xc_2, Fp, p->b.Wt_lo, Fp, (T & X3T_ENC_XLO) ? Fp : 0}};
            PROF(PS_ENC_GEMM, launch_gemm_f32out_n(dt, Bp, Hp, es3, 3, p->slabs, Hp, p->s_enc3, slab, st, GEMM_ROLE_ENCODE,
                                                   label_with_encode ? &lj : nullptr, label_with_encode ? &enc_label_done : nullptr));
        } else if (use_bits)
            PROF(PS_ENC_GEMM, launch_encode_bits(Bp, Hp, Fp, p->xc_bits, Fp / 32, p->b.Wt_lo, Fp, p->slabs, Hp, p->s_enc, slab, st,
                                                 label_with_encode ? &lj : nullptr, label_with_encode ? &enc_label_done : nullptr));
        else
            PROF(PS_ENC_GEMM, launch_gemm_f32out(dt, Bp, Hp, p->xc, Fp, p->b.Wt_lo, Fp, Fp, nullptr, 0, nullptr, 0, 0, p->slabs, Hp, p->s_enc, slab, st,
                                                 GEMM_ROLE_ENCODE, label_with_encode ? &lj : nullptr, label_with_encode ? &enc_label_done : nullptr));
        PROF(PS_ENC_FIN, launch_encode_finish(p->slabs, (x3 && dense_in) ? p->s_enc3 : p->s_enc, slab, Hp, p->b.bh, B, H, c.enc_act, dt, p->h_f32, p->h_lo, Hp,
                                              p->h_t, ldB, p->gram_split ? p->hcat_a : nullptr, p->gram_split ? p->hcat_b : nullptr,
                                              (T & (X3T_DH_HLO | X3T_DW_HLO)) ? p->h_t2 : nullptr, stream));
        labels_done = (label_in_gather && !s->c_indptr) || enc_label_done;
    }
    if (h_only) return 0;
    // 5-6. miners (K5-K7)
    const int Bt = explicit3 ? B / 3 : B;
    if (ext_mine) {
        // mined by the caller
    } else if (explicit3) {
        PROF(PS_LABEL, dae_label_stats(nullptr, Bt, Bp, DAE_TRIPLET_NONE, nullptr, nullptr, nullptr, nullptr, p->cw, 0.f, nullptr, stream));
        // every one of the 3*Bt stacked rows carries weight 1/(Bt + 1e-16): three unweighted row means (:303-305)
        DAE_CHECK_HIP(hipMemcpyAsync(p->cw + Bt, p->cw, (size_t)Bt * 4, hipMemcpyDeviceToDevice, st));
        DAE_CHECK_HIP(hipMemcpyAsync(p->cw + 2 * Bt, p->cw, (size_t)Bt * 4, hipMemcpyDeviceToDevice, st));
        PROF(PS_MINER, dae_explicit_triplet(p->h_f32, Hp, Bt, H, c.alpha, p->dh_extra, p->loss_part, p->tri_scalars, stream));
    } else if (!labels_done) {
        PROF(PS_LABEL, dae_label_stats(s->labels, B, Bp, c.triplet, p->n_same, p->acc, p->nvalid, p->dw_i64, p->cw, c.alpha, p->tri_scalars, stream));
    }
    bool forked = false, sym_ride = false;
    const bool fold_finalize = (c.triplet == DAE_TRIPLET_BATCH_ALL && !c.pos_triplets_only) && !ext_mine;
    // 7. decode + reconstruction loss + d cost/d z2   (K3/K4) -- on stream `ds`; `ride`: the launch also scales G + G^T (sym_scale)
    const int dbn = plan_dec_bn(p);
    const int ncw = 2 * Fp / dbn;
    auto decode_section = [&](hipStream_t ds, bool ride) -> int {
        DecodeEpi e;
        memset(&e, 0, sizeof(e));
        e.bv = p->b.bv; e.x = p->x; e.ldx = Fp; e.x_bits = use_xbits ? p->x_bits : nullptr; e.ldxb = Fp / 32; e.cw = p->cw; e.cos_stats = is_cos ? p->cos_stats : nullptr;
        e.rowloss_part = is_cos ? p->rowloss_part : nullptr; e.tile_part = is_cos ? nullptr : p->tile_part;
        e.dbv_part = backward ? p->dbv_part : nullptr; e.cos_part = p->cos_part;
        e.delta2 = backward ? p->delta2 : nullptr; e.ldd = Fp; e.delta2_t = (backward && !dw_tr) ? p->delta2_t : nullptr; e.lddt = ldB;
        e.B = B; e.F = F; e.Bp = Bp; e.Fp = Fp; e.dec_act = c.dec_act; e.loss_func = c.loss_func; e.ce_literal = p->ce_literal ? 1 : 0;
        e.op_scale = osc; e.bn = dbn;
        if (ride) { e.sym_G = p->G; e.sym_scalars = p->tri_scalars; e.sym_Gs = p->Gs; e.sym_B = B; e.sym_Bp = Bp; }
        // z2 = h W^T: one K segment, or -- split-bf16 -- (h_hi, W_hi) (h_hi, W_lo) (h_lo, W_hi); the row-major h_hi / h_lo are the first and
        // third block of the Gram operand hcat_a = [hi | hi | lo] (leading dimension 3 Hp)
        GemmSegDesc dsegs[3] = {{p->h_lo, Hp, p->b.W_lo, Hp, Hp}, {nullptr, 0, nullptr, 0, 0}, {nullptr, 0, nullptr, 0, 0}};
        int ndseg = 1;
        if (x3) {
            dsegs[0] = {p->hcat_a, 3 * (int64_t)Hp, p->b.W_lo, Hp, Hp};
            dsegs[1] = {p->hcat_a, 3 * (int64_t)Hp, p->W_lo2, Hp, (T & X3T_DEC_WLO) ? Hp : 0};      // K = 0: the term is dropped
            dsegs[2] = {p->hcat_a + (size_t)2 * Hp * 2, 3 * (int64_t)Hp, p->b.W_lo, Hp, (T & X3T_DEC_HLO) ? Hp : 0};
            ndseg = 3;
            if (backward) { e.delta2_2 = (T & X3T_DH_D2LO) ? p->delta2_2 : nullptr; e.delta2_t2 = (T & X3T_DW_D2LO) ? p->delta2_t2 : nullptr; }
            if (x2_clean) e.x2 = p->x_2;    // valued clean rows: x = hi + lo (RES instantiations)
        }
        if (is_cos) {
            e.cos_pass = 1;
            const bool zs = backward && p->cos_zstore_ok && p->zbuf;
            if (zs) { e.z_io = p->zbuf; e.ldz = Fp; e.z_mode = 1; }
            PROF(PS_DECODE, launch_decode_loss_n(dt, Bp, Fp, dsegs, ndseg, e, ds));
            if (zs) e.z_mode = 2;
            PROF(PS_COS_REDUCE, dae_cos_reduce(p->cos_part, ncw, B, Bp, p->cos_stats, p->rowloss_part, (void*)ds));
            e.sym_G = nullptr;                                 // the first pass carried the rider
            if (backward) { e.cos_pass = 2; PROF(PS_DECODE, launch_decode_loss_n(dt, Bp, Fp, dsegs, ndseg, e, ds)); }
        } else {
            e.cos_pass = 0;
            PROF(PS_DECODE, launch_decode_loss_n(dt, Bp, Fp, dsegs, ndseg, e, ds));
        }
        return 0;
    };
    if (!ext_mine && (c.triplet == DAE_TRIPLET_BATCH_ALL || c.triplet == DAE_TRIPLET_BATCH_HARD)) {
        const int64_t dslab = (int64_t)Bp * Bp;
        // the label block of this step's encode launch also ranked the anchors by sweep cost (only then is the buffer current)
        const int32_t* order = (labels_done && !ext_mine && p->miner_order_ok && c.triplet == DAE_TRIPLET_BATCH_ALL) ? p->miner_order : nullptr;
        const int32_t* cls = (labels_done && !ext_mine && p->miner_ranges_ok && c.triplet == DAE_TRIPLET_BATCH_ALL) ? p->cls_range : nullptr;
        // batch_all (all valid triplets): cw comes from the labels alone -> the Gram -> miner chain (VALU work, the longer of the two) and
        // the decode kernel (MFMA + loss epilogue) are independent until dL/dh.  Option "overlap": the decode forks onto the side stream,
        // the chain stays on the step's stream and joins before the dh GEMM (never while profiling: the slot events are per stream)
        const bool overlap = p->overlap_ok && !p->prof && c.triplet == DAE_TRIPLET_BATCH_ALL && !c.pos_triplets_only;
        const bool gram_first = overlap && p->overlap_mode >= 2;
        if (gram_first) PROF(PS_GRAM, launch_gram(p, Bp, Hp, dslab, st));
        if (overlap) {
            if (!p->side) {
                DAE_CHECK_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
                DAE_CHECK_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
                DAE_CHECK_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
            }
            DAE_CHECK_HIP(hipEventRecord(p->ev_fork, st));
            DAE_CHECK_HIP(hipStreamWaitEvent(p->side, p->ev_fork, 0));
            if (p->overlap_mode != 3) {
                RC(decode_section(p->side, false));
                DAE_CHECK_HIP(hipEventRecord(p->ev_join, p->side));
            }
            forked = true;
        }
        if (!gram_first) PROF(PS_GRAM, launch_gram(p, Bp, Hp, dslab, st));
        if (c.triplet == DAE_TRIPLET_BATCH_ALL)
            PROF(PS_MINER, launch_batch_all(p->D_slabs, p->s_gram, dslab, Bp, s->labels, B, Bp, 0, B,
                                     (c.pos_triplets_only ? DAE_MINER_POS_ONLY : 0) | (dt == DAE_BF16 ? DAE_MINER_FAST : 0), p->loss_part,
                                     p->cnt_part, p->G, p->role_cnt, order, st, cls));
        else
            PROF(PS_MINER, dae_triplet_batch_hard(p->D_slabs, p->s_gram, dslab, Bp, s->labels, B, Bp, p->loss_part, p->cnt_part, p->dw_i32, p->G,
                                      stream));
        if (overlap && p->overlap_mode == 3) {                // the miner's workgroups are queued first, the decode's fill in beside / behind them
            RC(decode_section(p->side, false));
            DAE_CHECK_HIP(hipEventRecord(p->ev_join, p->side));
        }
        if (!fold_finalize)   // batch_all over all valid triplets: scale comes from label_stats, sums from step_stats
            PROF(PS_TRI_FIN, dae_triplet_finalize(c.triplet, c.pos_triplets_only, B, Bp, c.alpha, p->loss_part, p->cnt_part, p->nvalid,
                                    p->dw_i32, p->role_cnt, p->dw_f32, p->cw, p->tri_scalars, stream));
        sym_ride = backward && p->sym_ride_ok && !forked;        // the decode launch below carries it
        if (backward && !sym_ride) PROF(PS_SYM, launch_sym_scale(p->G, B, Bp, p->tri_scalars, dt, p->Gs, osc, st));
    }
    if (forked) DAE_CHECK_HIP(hipStreamWaitEvent(st, p->ev_join, 0));   // join: delta2 and the loss partials are ready
    else RC(decode_section(st, sym_ride));
    // 8. statistics of this step (autoencoder.py:233 fetch list)
    StatsArgs sa{is_cos ? p->rowloss_part : nullptr, 1, is_cos ? nullptr : p->tile_part, (Bp / 128) * (Fp / dbn), p->cw, B, Bp,
                 c.triplet == 3 ? DAE_TRIPLET_BATCH_HARD : c.triplet, c.alpha, p->tri_scalars,
                 (c.triplet == DAE_TRIPLET_BATCH_ALL && !ext_mine) ? p->nvalid : nullptr, s->stats, fold_finalize ? p->loss_part : nullptr,
                 fold_finalize ? p->cnt_part : nullptr};
    const bool stats_in_tail = backward && tail;     // rides on the step-tail launch after the dW GEMM
    if (!stats_in_tail)
        PROF(PS_STATS, dae_step_stats(sa.rowloss_part, sa.n_col_waves, sa.tile_part, sa.n_tiles, sa.cw, B, Bp, sa.triplet, sa.alpha, sa.tri_scalars,
                                      sa.nvalid, sa.loss_part, sa.cnt_part, s->stats, stream));
    if (!backward) return 0;
    // 9-10. dL/dh = delta2 W + alpha (G+G^T) h ; delta1                     (K8)
    const bool mined = !ext_mine && (c.triplet == DAE_TRIPLET_BATCH_ALL || c.triplet == DAE_TRIPLET_BATCH_HARD);
    const int s_dh = (x3 && dense_in) ? p->s_dh3 : p->s_dh;
    if (x3) {       // (d2_hi, Wt_hi) (d2_hi, Wt_lo) (d2_lo, Wt_hi) + Gs.h_hi (+ Gs.h_lo with option x3_dh_hlo): Gs itself stays bf16 (tools/precision_study.py)
        const GemmSegDesc hs[5] = {{p->delta2, Fp, p->b.Wt_lo, Fp, Fp}, {p->delta2, Fp, p->Wt_lo2, Fp, (T & X3T_DH_WLO) ? Fp : 0},
                                   {p->delta2_2, Fp, p->b.Wt_lo, Fp, (T & X3T_DH_D2LO) ? Fp : 0},
                                   {p->Gs, Bp, p->h_t, ldB, mined ? Bk : 0}, {p->Gs, Bp, p->h_t2, ldB, (mined && (T & X3T_DH_HLO)) ? Bk : 0}};
        PROF(PS_DH_GEMM, launch_gemm_f32out_n(dt, Bp, Hp, hs, 5, p->slabs, Hp, s_dh, slab, st, GEMM_ROLE_DH, nullptr, nullptr, 1.f, B));
    } else {
        PROF(PS_DH_GEMM, launch_gemm_f32out(dt, Bp, Hp, p->delta2, Fp, p->b.Wt_lo, Fp, Fp, mined ? p->Gs : nullptr, Bp, mined ? p->h_t : nullptr, ldB,
                              mined ? Bk : 0, p->slabs, Hp, p->s_dh, slab, st, GEMM_ROLE_DH, nullptr, nullptr, B));
    }
    PROF(PS_DH_FIN, launch_dh_finish(p->slabs, s_dh, slab, Hp, (explicit3 || ext_mine) ? p->dh_extra : nullptr, p->h_f32, Hp, p->b.bh, B, H, c.enc_act, dt,
                     p->delta1_t, ldB, p->colsum_part, nullptr, nullptr, st, (T & X3T_DW_D1LO) ? p->delta1_t2 : nullptr, oinv, osc));
    // 11. dW = x~^T delta1 + delta2^T h                                      (K8, tied weights)
    // split-bf16 segment list (K = 0 segments are skipped): the third one exists only when x~^T has a lo image
    // (dw_tr: the A operands are the row-major images -- x~ from the CSR scatter (p->xct used as [Bp x Fp]) or the dense gather (p->xc), and delta2)
    const void* a_x = dw_tr ? (const void*)(dense_in ? p->xc : p->xct) : (const void*)p->xct;
    const void* a_d2 = dw_tr ? (const void*)p->delta2 : (const void*)p->delta2_t;
    const int64_t lda_w = dw_tr ? Fp : ldB;
    const GemmSegDesc ws3[6] = {{a_x, lda_w, p->delta1_t, ldB, Bk}, {p->xct, ldB, p->delta1_t2, ldB, (T & X3T_DW_D1LO) ? Bk : 0},
                                {p->xct_2, ldB, p->delta1_t, ldB, x3_vals ? Bk : 0},
                                {a_d2, lda_w, p->h_t, ldB, Bk}, {p->delta2_t, ldB, p->h_t2, ldB, (T & X3T_DW_HLO) ? Bk : 0},
                                {p->delta2_t2, ldB, p->h_t, ldB, (T & X3T_DW_D2LO) ? Bk : 0}};
    if (fuse_opt || dw_pc_grad) {
        OptEpi oe;
        memset(&oe, 0, sizeof(oe));
        oe.ldw = Hp; oe.ldwt = Fp; oe.gin = oinv;
        if (fuse_opt) {
            oe.W = p->b.W; oe.grad = s->phase == 3 ? nullptr : p->b.grad; oe.s1 = p->b.opt_s1; oe.s2 = p->b.opt_s2;
            oe.W_lo = p->b.W_lo; oe.Wt_lo = p->b.Wt_lo; oe.opt = c.opt; oe.lr = plan_lr(p, s->adam_t);
            oe.mom = c.momentum; oe.gscale = s->grad_scale;
        } else {                                     // data parallel: gradient to memory (fp32 flat buffer, or the bf16 exchange image)
            oe.opt = DW_OPT_GRAD_ONLY; oe.grad = p->b.grad_lo ? nullptr : p->b.grad; oe.grad_lo = p->b.grad_lo;
            oe.ldw = Hp;
        }
        if (x3) {   // x~^T.(d1_hi + d1_lo) + (d2^T_hi, h^T_hi) (d2^T_hi, h^T_lo) (d2^T_lo, h^T_hi); the epilogue writes both parts of both shadows
            oe.W_lo2 = (T & X3T_DEC_WLO) ? p->W_lo2 : nullptr; oe.Wt_lo2 = p->Wt_lo2;     // W_lo2 feeds the decode's (h_hi, W_lo) term only
            PROF(PS_DW_GEMM, launch_dw_opt_n(Fp, Hp, ws3, 6, oe, st, p->dw_pair_ok, dw_tr));
        } else if (dw_bits) {
            DwBitsArgs xa{p->xtb, ldB / 32, s->scale};
            PROF(PS_DW_GEMM, launch_dw_opt(Fp, Hp, nullptr, ldB, p->delta1_t, ldB, Bk, p->delta2_t, ldB, p->h_t, ldB, Bk, oe, st, &xa));
        } else {
            PROF(PS_DW_GEMM, launch_dw_opt(Fp, Hp, a_x, lda_w, p->delta1_t, ldB, Bk, a_d2, lda_w, p->h_t, ldB, Bk, oe, st, nullptr, dw_tr));
        }
    } else if (x3) {
        PROF(PS_DW_GEMM, launch_gemm_f32out_n(dt, Fp, Hp, ws3, 6, p->b.grad, Hp, 1, 0, st, GEMM_ROLE_DW, nullptr, nullptr, oinv));
    } else {
        const GemmSegDesc ws2[2] = {{p->xct, ldB, p->delta1_t, ldB, Bk}, {p->delta2_t, ldB, p->h_t, ldB, Bk}};
        PROF(PS_DW_GEMM, launch_gemm_f32out_n(dt, Fp, Hp, ws2, 2, p->b.grad, Hp, 1, 0, st, GEMM_ROLE_DW, nullptr, nullptr, oinv));
        // data parallel with a bf16 exchange image: the shape did not fit the kernel that writes it directly
        if (!apply_now && dt == DAE_BF16 && p->b.grad_lo) RC(launch_cast_bf16(p->b.grad, p->b.grad_lo, (int64_t)Fp * Hp, st));
    }
    p->ev_dw_live = false;
    if (p->ev_dw) { DAE_CHECK_HIP(hipEventRecord(p->ev_dw, st)); p->ev_dw_live = true; }      // the W gradient (grad / grad_lo) is complete from here on
    // 12. bias gradients
    float* g_bh = p->b.grad + (int64_t)Fp * Hp;
    const int64_t boff = (int64_t)Fp * Hp;
    const bool fuse_bias = apply_now;               // single-GPU step: the bias update rides on the bias-gradient kernel
    if (tail) {
        BiasArgs ba{p->dbv_part, 2 * Bp / 128, p->colsum_part, Bp / 32, p->b.bh, H, Hp, F, Fp, c.enc_act, g_bh, g_bh + Hp,
                    fuse_bias ? 1 : 0, c.opt, plan_lr(p, s->adam_t), c.momentum, s->grad_scale, p->b.bv,
                    p->b.opt_s1 ? p->b.opt_s1 + boff : nullptr, p->b.opt_s2 ? p->b.opt_s2 + boff : nullptr};
        ClearArgs ca{s->c_indptr ? s->c_indptr : p->b.indptr, s->c_indptr ? s->c_indices : p->b.indices,
                     (s->c_indptr && s->c_row_idx) ? s->c_row_idx : s->row_idx, B, F, dw_bits ? nullptr : p->xct, dw_tr ? (int64_t)Fp : (int64_t)ldB, p->es,
                     dw_bits ? p->xtb : nullptr, ldB / 32, x3_vals ? p->xct_2 : nullptr, dw_tr ? 1 : 0};
        PROF(PS_BIAS, launch_step_tail(ba, &sa, csr_in ? &ca : nullptr, st));
        if (csr_in) { if (dw_bits) p->xtb_clean = true; else p->xct_clean = true; if (x3_vals) p->xct2_clean = true; }
    } else {
        PROF(PS_BIAS, dae_bias_grads(p->dbv_part, 2 * Bp / 128, p->colsum_part, Bp / 32, p->b.bh, H, Hp, F, Fp, c.enc_act, g_bh, g_bh + Hp,
                                     fuse_bias ? 1 : 0, c.opt, plan_lr(p, s->adam_t), c.momentum, s->grad_scale, p->b.bv,
                                     p->b.opt_s1 ? p->b.opt_s1 + boff : nullptr, p->b.opt_s2 ? p->b.opt_s2 + boff : nullptr, stream));
    }
    if (s->phase == 1 || s->phase == 5 || fuse_opt) return 0;
    // 13. optimizer (K9): W (+ shadows); biases were updated above
    PROF(PS_OPT, launch_opt_step(c.opt, plan_lr(p, s->adam_t), c.momentum, s->grad_scale, p->b.W, p->b.bh, p->b.bv, p->b.grad, p->b.opt_s1,
                                 p->b.opt_s2, Fp, Hp, dt, p->b.W_lo, p->b.Wt_lo, plan_w_lo2(p), x3 ? p->Wt_lo2 : nullptr, /*apply=*/2, stream));
    return 0;
}

extern "C" int dae_plan_apply(dae_plan* p, int32_t adam_t, float grad_scale, void* stream) {
    DAE_CHECK_ARG(p && p->bound, "plan_apply: plan not bound");
    const float lr = plan_lr(p, adam_t);
    return launch_opt_step(p->cfg.opt, lr, p->cfg.momentum, grad_scale, p->b.W, p->b.bh, p->b.bv, p->b.grad, p->b.opt_s1, p->b.opt_s2,
                           p->Fp, p->Hp, p->cfg.dtype, p->b.W_lo, p->b.Wt_lo, plan_w_lo2(p), p->x3 ? p->Wt_lo2 : nullptr, /*apply=*/1, stream);
}

// The same on the row band [f0, f1) of W (multiples of 64) -- dp.AllReduceExchange with buckets: the flat gradient is all-reduced band by band and every
// band is applied as soon as its sum has arrived, while the next band is still on the wire.  The band that ends at Fp also updates the biases (their
// gradients sit behind the W part of the flat buffer, i.e. in the last bucket).  Bands applied in any order give the weights of one dae_plan_apply.
extern "C" int dae_plan_apply_band(dae_plan* p, int32_t adam_t, float grad_scale, int32_t f0, int32_t f1, void* stream) {
    DAE_CHECK_ARG(p && p->bound, "plan_apply_band: plan not bound");
    const float lr = plan_lr(p, adam_t);
    return launch_opt_step(p->cfg.opt, lr, p->cfg.momentum, grad_scale, p->b.W, p->b.bh, p->b.bv, p->b.grad, p->b.opt_s1, p->b.opt_s2,
                           p->Fp, p->Hp, p->cfg.dtype, p->b.W_lo, p->b.Wt_lo, plan_w_lo2(p), p->x3 ? p->Wt_lo2 : nullptr, /*apply=*/1, stream, f0, f1);
}

// ---- the exchange itself, on the step's stream (dae_comm.hip holds the communicator; reference step: autoencoder.py:206-246) ----
namespace dae {
int comm_allreduce_sum(dae_comm* c, float* buf, int64_t n, hipStream_t st);
hipStream_t comm_wire(dae_comm* c);
hipEvent_t comm_ev_ready(dae_comm* c);
hipEvent_t comm_ev_band(dae_comm* c, int k);
}  // namespace dae

// In-place all-reduce(sum) of the plan's flat fp32 gradient [dW (Fp x Hp) | dbh (Hp) | dbv (Fp)] over the communicator, enqueued on `stream` --
// the stream the step's kernels run on, so it starts when the gradient is complete and dae_plan_apply behind it needs no cross-stream wait.
extern "C" int dae_allreduce_grads(dae_plan* p, dae_comm* c, void* stream) {
    DAE_CHECK_ARG(p && p->bound && c, "allreduce_grads: plan not bound / null communicator");
    DAE_CHECK_ARG(!p->b.grad_lo, "allreduce_grads: the plan writes a 16-bit exchange image (grad_lo); the flat fp32 gradient would be stale");
    return comm_allreduce_sum(c, p->b.grad, (int64_t)p->Fp * p->Hp + p->Hp + p->Fp, (hipStream_t)stream);
}

// The whole second half of a data-parallel step after dae_train_step(phase = 1): all-reduce of the flat gradient + the optimizer on every rank.
//   buckets <= 1: ncclAllReduce and dae_plan_apply back to back on `stream`.
//   buckets  > 1: the flat buffer is cut into row bands of W (dae_dp_bands).  The collectives run on the communicator's wire stream: the bands
//     that hold W rows only start behind the dW GEMM (the plan's ev_dw, i.e. beside the step's tail kernel), the last band -- it carries the bias
//     gradients the tail writes -- behind the tail.  `stream` applies band k (dae_plan_apply_band) as soon as band k has been reduced, while band
//     k + 1 is still on the wire.  Element-wise the same sums and the same update as one bucket.
// Costs per step on the host: 2 event records + (buckets + 2) stream waits -- microseconds, where torch.distributed took ~25 us per collective.
extern "C" int dae_dp_exchange(dae_plan* p, dae_comm* c, int32_t adam_t, float grad_scale, int32_t buckets, void* stream) {
    DAE_CHECK_ARG(p && p->bound && c, "dp_exchange: plan not bound / null communicator");
    DAE_CHECK_ARG(!p->b.grad_lo, "dp_exchange: the plan writes a 16-bit exchange image (grad_lo); the flat fp32 gradient would be stale");
    hipStream_t st = (hipStream_t)stream;
    const int Fp = p->Fp, Hp = p->Hp;
    const int64_t n_flat = (int64_t)Fp * Hp + Hp + Fp;
    int32_t bounds[DAE_COMM_MAX_BUCKETS + 1];
    const int nb = dae_dp_bands(Fp, buckets, bounds);
    if (nb == 1) {
        RC(comm_allreduce_sum(c, p->b.grad, n_flat, st));
        return dae_plan_apply(p, adam_t, grad_scale, stream);
    }
    hipStream_t wire = comm_wire(c);
    if (!p->ev_dw) DAE_CHECK_HIP(hipEventCreateWithFlags(&p->ev_dw, hipEventDisableTiming));      // recorded by the steps enqueued from now on
    if (p->ev_dw_live) {
        DAE_CHECK_HIP(hipStreamWaitEvent(wire, p->ev_dw, 0));                 // W gradient complete: bands 0 .. nb-2 run beside the step's tail
    } else {
        DAE_CHECK_HIP(hipEventRecord(comm_ev_ready(c), st));
        DAE_CHECK_HIP(hipStreamWaitEvent(wire, comm_ev_ready(c), 0));
    }
    for (int k = 0; k < nb; ++k) {
        const int64_t lo = (int64_t)bounds[k] * Hp, hi = k + 1 < nb ? (int64_t)bounds[k + 1] * Hp : n_flat;
        if (k == nb - 1 && p->ev_dw_live) {                                   // the bias gradients sit behind the W part: wait for the tail too
            DAE_CHECK_HIP(hipEventRecord(comm_ev_ready(c), st));
            DAE_CHECK_HIP(hipStreamWaitEvent(wire, comm_ev_ready(c), 0));
        }
        RC(comm_allreduce_sum(c, p->b.grad + lo, hi - lo, wire));
        DAE_CHECK_HIP(hipEventRecord(comm_ev_band(c, k), wire));
    }
    for (int k = 0; k < nb; ++k) {
        DAE_CHECK_HIP(hipStreamWaitEvent(st, comm_ev_band(c, k), 0));
        RC(dae_plan_apply_band(p, adam_t, grad_scale, bounds[k], bounds[k + 1], stream));
    }
    return 0;
}

// Data-parallel second half with a SHARDED optimizer: this rank owns rows [f0, f1) of W.  grad_rows holds the
// rank-summed gradient of those rows (fp32 [f1-f0 x Hp], the output of the reduce-scatter); biases are updated on every rank
// from the all-reduced bias part of the plan's flat gradient when update_bias != 0.  Afterwards the ranks all-gather W_lo and
// call dae_plan_refresh_wt.
extern "C" int dae_plan_apply_rows(dae_plan* p, int32_t adam_t, float grad_scale, const float* grad_rows, int32_t f0, int32_t f1,
                                   int32_t update_bias, void* stream) {
    // split-bf16 mode: only the fp32 master rows (and their hi image) are current afterwards; the caller all-gathers the MASTER rows and
    // rebuilds all four shadows with dae_plan_sync_shadows (dp.ShardedExchange)
    DAE_CHECK_ARG(p && p->bound && grad_rows, "plan_apply_rows: plan not bound / null gradient");
    DAE_CHECK_ARG(f0 >= 0 && f0 <= f1 && f1 <= p->Fp && f0 % 64 == 0 && f1 % 64 == 0, "plan_apply_rows: rows [%d, %d) outside [0, %d] or not multiples of 64", f0, f1, p->Fp);
    const float lr = plan_lr(p, adam_t);
    RC(dae_opt_step_rows(p->cfg.opt, lr, p->cfg.momentum, grad_scale, p->b.W, grad_rows, p->b.opt_s1, p->b.opt_s2, p->Hp, f0, f1, p->cfg.dtype,
                         p->b.W_lo, stream));
    if (update_bias) {
        const int64_t off = (int64_t)p->Fp * p->Hp;
        dim3 grid((p->Hp + p->Fp + 255) / 256), block(256);
        RC(dae_opt_bias(p->cfg.opt, lr, p->cfg.momentum, grad_scale, p->b.bh, p->b.bv, p->b.grad + off, p->b.opt_s1 ? p->b.opt_s1 + off : nullptr,
                        p->b.opt_s2 ? p->b.opt_s2 + off : nullptr, p->Hp, p->Fp, stream));
    }
    return 0;
}

// Data parallel: make `stream` wait until the W gradient of the LAST enqueued dae_train_step is complete (the event sits between the dW
// GEMM and the step's tail kernel).  The first call creates the event; steps enqueued before it are not covered (returns DAE_WAIT_DW_CREATED then;
// 0 = the wait was enqueued; any other value is an error, text in dae_last_error).
extern "C" int dae_plan_stream_wait_dw(dae_plan* p, void* stream) {
    DAE_CHECK_ARG(p && p->bound, "plan_stream_wait_dw: plan not bound");
    if (!p->ev_dw) {
        DAE_CHECK_HIP(hipEventCreateWithFlags(&p->ev_dw, hipEventDisableTiming));
        return DAE_WAIT_DW_CREATED;      // not an error (dae_last_error untouched): the event covers the steps enqueued from now on
    }
    DAE_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, p->ev_dw, 0));
    return 0;
}

// Packed form of the sharded second half (dp.ShardedExchange): the low-precision rows of [f0, f1) go straight into the all-gather SEND
// buffer (row f at send + (f - f0) * Hp * es) and this rank's bias gradients are copied behind them (bias_off_bytes) -- the biases ride on
// the all-gather instead of their own all-reduce, and no copy of the updated rows is needed.  Biases are NOT updated here.
extern "C" int dae_plan_apply_rows_packed(dae_plan* p, int32_t adam_t, float grad_scale, const float* grad_rows, int32_t f0, int32_t f1,
                                          void* send, int64_t bias_off_bytes, void* stream) {
    DAE_CHECK_ARG(!p || !p->x3, "plan_apply_rows_packed: split-bf16 mode has no data-parallel path yet");
    DAE_CHECK_ARG(p && p->bound && grad_rows && send, "plan_apply_rows_packed: plan not bound / null buffer");
    DAE_CHECK_ARG(f0 >= 0 && f0 <= f1 && f1 <= p->Fp && f0 % 64 == 0 && f1 % 64 == 0, "plan_apply_rows_packed: rows [%d, %d) outside [0, %d] or not multiples of 64", f0, f1, p->Fp);
    const float lr = plan_lr(p, adam_t);
    char* lo_base = (char*)send - (int64_t)f0 * p->Hp * p->es;                 // dae_opt_step_rows writes row f at base + f * Hp * es
    RC(dae_opt_step_rows(p->cfg.opt, lr, p->cfg.momentum, grad_scale, p->b.W, grad_rows, p->b.opt_s1, p->b.opt_s2, p->Hp, f0, f1, p->cfg.dtype,
                         lo_base, stream));
    DAE_CHECK_HIP(hipMemcpyAsync((char*)send + bias_off_bytes, p->b.grad + (int64_t)p->Fp * p->Hp, (size_t)(p->Hp + p->Fp) * sizeof(float),
                                 hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// After the all-gather of the packed chunks: W_lo, Wt_lo and the biases of this rank (see dp_unpack_kernel).
extern "C" int dae_plan_dp_unpack(dae_plan* p, const void* recv, int32_t world, int32_t chunk_rows, int64_t chunk_stride_bytes,
                                  int64_t bias_off_bytes, int32_t adam_t, float grad_scale, void* stream) {
    DAE_CHECK_ARG(!p || !p->x3, "plan_dp_unpack: split-bf16 mode has no data-parallel path yet");
    DAE_CHECK_ARG(p && p->bound && recv, "plan_dp_unpack: plan not bound / null buffer");
    const int64_t off = (int64_t)p->Fp * p->Hp;
    return dae_dp_unpack(recv, world, chunk_rows, chunk_stride_bytes, bias_off_bytes, p->Fp, p->Hp, p->cfg.dtype, p->b.W_lo, p->b.Wt_lo, p->cfg.opt,
                         plan_lr(p, adam_t), p->cfg.momentum, grad_scale, p->b.bh, p->b.bv, p->b.opt_s1 ? p->b.opt_s1 + off : nullptr,
                         p->b.opt_s2 ? p->b.opt_s2 + off : nullptr, p->b.grad + off, stream);
}

extern "C" int dae_plan_refresh_wt(dae_plan* p, void* stream) {
    DAE_CHECK_ARG(!p || !p->x3, "plan_refresh_wt: split-bf16 mode has no data-parallel path yet");
    DAE_CHECK_ARG(p && p->bound, "plan_refresh_wt: plan not bound");
    return dae_transpose_shadow(p->b.W_lo, p->Fp, p->Hp, p->cfg.dtype, p->b.Wt_lo, stream);
}

extern "C" int dae_encode_rows(dae_plan* p, const int32_t* row_idx, int32_t B, float scale, const int64_t* indptr,
                               const int32_t* indices, const float* values, const float* dense, int64_t ld_dense, float* out,
                               int64_t ld_out, void* stream) {
    DAE_CHECK_ARG(p && p->bound && row_idx && out, "encode_rows: bad arguments");
    DAE_CHECK_ARG(B > 0 && B <= p->Bmax, "encode_rows: batch %d outside (0, %d]", B, p->Bmax);
    DAE_CHECK_ARG((indptr != nullptr) != (dense != nullptr), "encode_rows: give either a CSR or a dense matrix");
    hipStream_t st = (hipStream_t)stream;
    const int Bp = (int)pad128(B), Fp = p->Fp, Hp = p->Hp, dt = p->cfg.dtype;
    if (indptr && p->sparse_ok) {        // CSR: one launch on the stored entries, straight into the caller's [B x H] matrix when it is padded like h
        EncCsrLaunch q;
        memset(&q, 0, sizeof(q));
        q.indptr = indptr; q.indices = indices; q.values = values; q.row_idx = row_idx; q.B = B; q.F = p->F; q.H = p->H; q.dtype = dt;
        const bool w32 = p->enc_w32_ok && dt == DAE_BF16;
        q.W = w32 ? (const void*)p->b.W : (const void*)p->b.W_lo; q.w_f32 = w32 ? 1 : 0; q.w32_cols = p->w32_cols;
        q.ldw = Hp; q.bh = p->b.bh; q.enc_act = p->cfg.enc_act; q.corr_mode = DAE_CORR_NONE; q.scale = scale;
        q.h_f32 = p->h_f32; q.ldh = Hp;
        RC(launch_encode_csr(q, st));
        DAE_CHECK_HIP(hipMemcpy2DAsync(out, (size_t)ld_out * 4, p->h_f32, (size_t)Hp * 4, (size_t)p->H * 4, B, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    const bool use_bits = p->bits_ok && indptr && !values && scale == 1.0f;
    RC(gather_batch(p, indptr, indices, values, dense, ld_dense, row_idx, B, nullptr, use_bits ? nullptr : p->xc, nullptr, nullptr,
                    DAE_CORR_NONE, nullptr, 0, 0, 0.f, scale, stream, use_bits ? p->xc_bits : nullptr));
    const int64_t slab = (int64_t)Bp * Hp;
    if (p->x3 && dense) {        // split mode: x = hi + lo against W^T = hi + lo (the lo terms the plan keeps), as the training step encodes
        const uint32_t T = p->terms;
        if (T & X3T_ENC_XLO) RC(gather_dense_lo(p, dense, ld_dense, row_idx, B, nullptr, p->xc_2, nullptr, DAE_CORR_NONE, nullptr, 0, 0, 0.f, scale, stream));
        const GemmSegDesc es3[3] = {{p->xc, Fp, p->b.Wt_lo, Fp, Fp}, {p->xc, Fp, p->Wt_lo2, Fp, (T & X3T_ENC_WLO) ? Fp : 0},
                                    {p->xc_2, Fp, p->b.Wt_lo, Fp, (T & X3T_ENC_XLO) ? Fp : 0}};
        RC(launch_gemm_f32out_n(dt, Bp, Hp, es3, 3, p->slabs, Hp, p->s_enc3, slab, st, GEMM_ROLE_ENCODE));
        RC(dae_encode_finish(p->slabs, p->s_enc3, slab, Hp, p->b.bh, B, p->H, p->cfg.enc_act, dt, p->h_f32, nullptr, Hp, nullptr, 0, nullptr, nullptr,
                             stream));
        DAE_CHECK_HIP(hipMemcpy2DAsync(out, (size_t)ld_out * 4, p->h_f32, (size_t)Hp * 4, (size_t)p->H * 4, B, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (use_bits)
        RC(launch_encode_bits(Bp, Hp, Fp, p->xc_bits, Fp / 32, p->b.Wt_lo, Fp, p->slabs, Hp, p->s_enc, slab, st));
    else
        RC(launch_gemm_f32out(dt, Bp, Hp, p->xc, Fp, p->b.Wt_lo, Fp, Fp, nullptr, 0, nullptr, 0, 0, p->slabs, Hp, p->s_enc, slab, st,
                              GEMM_ROLE_ENCODE));
    RC(dae_encode_finish(p->slabs, p->s_enc, slab, Hp, p->b.bh, B, p->H, p->cfg.enc_act, dt, p->h_f32, nullptr, Hp, nullptr, 0, nullptr, nullptr,
                         stream));
    DAE_CHECK_HIP(hipMemcpy2DAsync(out, (size_t)ld_out * 4, p->h_f32, (size_t)Hp * 4, (size_t)p->H * 4, B, hipMemcpyDeviceToDevice, st));
    return 0;
}
