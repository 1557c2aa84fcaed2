"""GPU parity tests of the fp16 build of the library (libdae_hip_f16.so: the same sources with the 16-bit storage format switched to IEEE
fp16 and v_mfma_f32_32x32x16_f16) -- precision 'f16x2' (fp16 operand images, W kept as hi + lo: two product terms in the decode and dh
GEMMs, one in dW), 'f16x3' (every operand hi + lo) and 'f16' (single images) -- against the fp64 CPU oracle on identical seeded inputs.

What is specific to this build and covered here: the power-of-two operand scale of the back-propagated images (delta2 / Gs / delta1 are
stored times op_scale and un-scaled by dh_finish and the dW epilogue -- every path that writes a gradient: fused optimizer, gradient-only
phase, un-fused GEMM + optimizer kernel, shapes beyond one dW round), the lo-term mask of the split mode, and the MFMA's treatment of
subnormal fp16 operands (the lo image of W lives there: |W - fp16(W)| <= 2^-12 |W| ~ 6e-6 < 6.1e-5)."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_hip_step import _rel, _run_case

pytestmark = pytest.mark.gpu


def test_f16_library_is_the_fp16_build():
    from dae_rnn_news_recommendation_amd import _lib as L
    assert L.load("f16").dae_storage_format() == 1 and L.load("bf16").dae_storage_format() == 0
    assert L.load("f16").dae_abi_version() == L.ABI_VERSION


def test_mfma_f16_keeps_subnormal_operands():
    """C = A.B^T through dae_gemm_nt of the fp16 build with A = 2^-20 (a subnormal fp16) and B = 2^10: every product is 2^-10 and
    C = K * 2^-10 exactly.  A matrix core that flushed subnormal inputs would return 0 -- and the (h, W_lo) product terms with it."""
    from dae_rnn_news_recommendation_amd import _lib as L
    lib = L.load("f16")
    M = N = 128; K = 256
    A = torch.full((M, K), 2.0 ** -20, dtype=torch.float16, device="cuda")
    B = torch.full((N, K), 2.0 ** 10, dtype=torch.float16, device="cuda")
    assert float(A[0, 0]) == 2.0 ** -20 and float(A[0, 0]) < 6.1e-5
    Cm = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    L.check(lib.dae_gemm_nt(L.BF16, M, N, L.ptr(A), K, L.ptr(B), K, K, None, 0, None, 0, 0, L.ptr(Cm), N, 1, 0, L.current_stream()), "dae_gemm_nt", lib)
    torch.cuda.synchronize()
    assert torch.all(Cm == K * 2.0 ** -10), (float(Cm.min()), float(Cm.max()), K * 2.0 ** -10)


def test_plan_defaults_of_the_f16_build():
    from dae_rnn_news_recommendation_amd.engine import Engine
    e = Engine(10000, 500, 800, dtype="f16x2", triplet="batch_all")
    i = e.info()
    assert i["storage"] == "f16" and i["split"] and i["x3_terms"] == (1 | 4) and i["op_scale"] == 8192.0, i
    assert e.td == torch.float16
    e3 = Engine(600, 64, 100, dtype="f16x3", triplet="batch_all")
    assert e3.info()["x3_terms"] == (1 << 11) - 1 and e3.info()["op_scale"] == 1024.0
    b = Engine(600, 64, 100, dtype="bf16x3", triplet="batch_all")
    assert b.info()["storage"] == "bf16" and b.info()["op_scale"] == 1.0 and b.info()["x3_terms"] == (1 << 11) - 1


@pytest.mark.parametrize("opt", ["gradient_descent", "adam"])
@pytest.mark.parametrize("strategy", ["none", "batch_all", "batch_hard"])
def test_step_f16x3_matches_oracle(strategy, opt):
    """Every operand hi + lo fp16 (22 significant bits): as close to the fp64 oracle as the split-bf16 mode (16 bits) -- the same gates as
    test_step_bf16x3_matches_oracle.  Exercises every lo image and the operand scale through the fused dW + optimizer epilogue."""
    out, ref, got = _run_case("f16x3", strategy, "cross_entropy", ("sigmoid", "sigmoid"), opt, steps=3)
    tol = 2e-5 if opt == "gradient_descent" else 1e-4
    for r, st, dW, dbh, dbv in out:
        assert abs(st[1] - r["ae_loss"]) <= tol * abs(r["ae_loss"]), (st[1], r["ae_loss"])
        assert abs(st[0] - r["cost"]) <= tol * abs(r["cost"])
        if strategy == "batch_all":
            assert abs(st[2] - r["triplet_loss"]) <= tol * abs(r["triplet_loss"])
        assert _rel(dW, r["dW"]) < 1e-4 and _rel(dbh, r["dbh"]) < 1e-4 and _rel(dbv, r["dbv"]) < 1e-4, (_rel(dW, r["dW"]), _rel(dbh, r["dbh"]))
    if opt == "gradient_descent":
        for a, b in zip(got, ref):
            assert _rel(a, b) < 2e-4, _rel(a, b)


@pytest.mark.parametrize("strategy,loss_func,acts", [("none", "cross_entropy", ("sigmoid", "sigmoid")), ("batch_all", "cross_entropy", ("sigmoid", "sigmoid")),
                                                     ("batch_hard", "cross_entropy", ("sigmoid", "sigmoid")), ("batch_all", "mean_squared", ("tanh", "none")),
                                                     ("none", "cosine_proximity", ("sigmoid", "sigmoid"))])
def test_step_f16x2_matches_oracle(strategy, loss_func, acts):
    """The product's parity mode: single fp16 images of h / delta2 / delta1 / Gs / x~^T (11 significant bits, random rounding) and W as hi + lo.
    One step's losses sit at the oracle (the forward pass multiplies h by a 22-bit W); its gradients carry the 2^-12 operand rounding --
    an order of magnitude closer than plain bf16 (2e-2)."""
    out, ref, got = _run_case("f16x2", strategy, loss_func, acts, "gradient_descent", steps=3)
    worst = 0.0
    for r, st, dW, dbh, dbv in out:
        assert abs(st[1] - r["ae_loss"]) <= 5e-5 * abs(r["ae_loss"]), (st[1], r["ae_loss"])
        assert abs(st[0] - r["cost"]) <= 5e-5 * abs(r["cost"])
        worst = max(worst, _rel(dW, r["dW"]), _rel(dbh, r["dbh"]), _rel(dbv, r["dbv"]))
    print(f"[f16x2] {strategy} {loss_func}: worst gradient deviation {worst:.2e}")
    assert worst < 1.5e-3, worst
    for a, b in zip(got, ref):
        assert _rel(a, b) < 2e-3, _rel(a, b)


@pytest.mark.parametrize("dtype", ["f16x2", "f16x3"])
def test_step_f16_gradient_only_phase_and_unfused_optimizer(dtype):
    """The operand scale must be divided out on EVERY path that produces a gradient: phase 1 (gradient to the flat buffer, dae_plan_apply
    afterwards) and fused_opt = 0 (N-segment GEMM to memory + optimizer kernel) against the fused step."""
    kw = dict(steps=3, seed=5)
    a, ra, pa = _run_case(dtype, "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), "momentum", **kw)
    b, rb, pb = _run_case(dtype, "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), "momentum", phase=1, **kw)
    c, rc, pc = _run_case(dtype, "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), "momentum", options={"fused_opt": 0}, **kw)
    for (_, sa, dWa, *_), (_, sb, dWb, *_), (_, sc, dWc, *_) in zip(a, b, c):
        assert np.allclose(sa[:3], sb[:3], rtol=1e-5, atol=0) and np.allclose(sa[:3], sc[:3], rtol=1e-5, atol=0), (sa, sb, sc)
        assert _rel(dWb, np.asarray(dWa, np.float64)) < 2e-5 and _rel(dWc, np.asarray(dWa, np.float64)) < 2e-5
    for u, v, w in zip(pa, pb, pc):
        assert _rel(v, np.asarray(u, np.float64)) < 2e-5 and _rel(w, np.asarray(u, np.float64)) < 2e-5
    gate = 1e-4 if dtype == "f16x3" else 1.5e-3
    for r, st, dW, dbh, dbv in b:
        assert _rel(dW, r["dW"]) < gate and _rel(dbh, r["dbh"]) < gate, (_rel(dW, r["dW"]), _rel(dbh, r["dbh"]))


@pytest.mark.parametrize("dtype", ["f16x2", "f16x3"])
def test_step_f16_dense_and_valued_input(dtype):
    """Dense ndarray train set (gather + encode GEMM with the W^T lo term) and valued CSR under a scale factor."""
    gate = 1e-4 if dtype == "f16x3" else 1.5e-3
    for kw in (dict(dense=True), dict(scale=0.7)):
        out, ref, got = _run_case(dtype, "batch_all", "mean_squared" if "dense" in kw else "cross_entropy",
                                  ("tanh", "none") if "dense" in kw else ("sigmoid", "sigmoid"), "gradient_descent", steps=3, **kw)
        for r, st, dW, dbh, dbv in out:
            assert abs(st[0] - r["cost"]) <= (2e-5 if dtype == "f16x3" else 1e-4) * abs(r["cost"]), (kw, st[0], r["cost"])
            assert _rel(dW, r["dW"]) < gate and _rel(dbh, r["dbh"]) < gate, (kw, _rel(dW, r["dW"]), _rel(dbh, r["dbh"]))
        for a, b in zip(got, ref):
            assert _rel(a, b) < 2 * gate, (kw, _rel(a, b))


def test_step_f16x2_shape_beyond_one_dw_round():
    """More 160 x 128 tiles than CUs: the dW GEMM goes to memory (un-scaled by its output scale) + the optimizer kernel."""
    out, ref, got = _run_case("f16x2", "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), "momentum", N=300, F=5000, H=1200, B=128, steps=2)
    for r, st, dW, dbh, dbv in out:
        assert abs(st[0] - r["cost"]) <= 5e-5 * abs(r["cost"]), (st[0], r["cost"])
        assert _rel(dW, r["dW"]) < 1.5e-3, _rel(dW, r["dW"])
    for a, b in zip(got, ref):
        assert _rel(a, b) < 2e-3, _rel(a, b)


def test_step_f16x2_op_scale_is_transparent():
    """A different power of two in the operand images changes nothing but the rounding of subnormal entries: the step at op_scale 2^6 and 2^13
    agrees to fp16 rounding, and both sit at the oracle."""
    a, ra, pa = _run_case("f16x2", "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), "gradient_descent", steps=2, seed=9, options={"op_scale_log2": 13})
    b, rb, pb = _run_case("f16x2", "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), "gradient_descent", steps=2, seed=9, options={"op_scale_log2": 6})
    for (_, sa, dWa, *_), (_, sb, dWb, *_) in zip(a, b):
        assert np.allclose(sa[:3], sb[:3], rtol=2e-5, atol=0)
        assert _rel(dWb, np.asarray(dWa, np.float64)) < 1e-3


@pytest.mark.parametrize("dtype,loss_func,acts", [("f16x2", "cross_entropy", ("sigmoid", "sigmoid")), ("bf16", "cosine_proximity", ("sigmoid", "sigmoid")),
                                                  ("f16", "mean_squared", ("tanh", "none"))])
def test_step_wide_decode_tile_equals_narrow(dtype, loss_func, acts):
    """Plan option decode_bn = 128 (the 128 x 128 decode tile the plan picks by itself for F = 50000) against the default 128 x 64 tile: every logit is
    the same K-ordered MFMA sum, so delta2, the gradients and the parameters agree to the last bit; only the loss partial sums are added in another order."""
    from dae_rnn_news_recommendation_amd import _lib as L
    kw = dict(steps=2, seed=17, N=400, F=900, H=90, B=150)
    try:        # (decode_x3 = 0: the narrow tile on the segment walk, the K order of the wide kernel; its register-carry loops interleave the hi / lo products per K tile)
        a, _, pa = _run_case(dtype, "batch_all", loss_func, acts, "gradient_descent", options={"decode_bn": 64, "decode_x3": 0}, **kw)
    finally:
        L.set_glds_all(-18)
    b, _, pb = _run_case(dtype, "batch_all", loss_func, acts, "gradient_descent", options={"decode_bn": 128}, **kw)
    for (_, sa, dWa, dbha, dbva), (_, sb, dWb, dbhb, dbvb) in zip(a, b):
        assert np.allclose(sa[:3], sb[:3], rtol=2e-6, atol=0), (sa, sb)
        if loss_func == "cosine_proximity":      # its row statistics (sum y^2, sum xhat.y) are partial sums over the tiles: another order, last-bit differences in delta2
            assert _rel(dWa, np.asarray(dWb, np.float64)) < 5e-4 and _rel(dbha, np.asarray(dbhb, np.float64)) < 5e-4      # (a flipped bf16 rounding of delta2 = 2^-9 of one element)
        else:
            assert np.array_equal(dWa, dWb) and np.array_equal(dbha, dbhb)
        assert np.allclose(dbva, dbvb, rtol=1e-5, atol=1e-9)
    for u, v in zip(pa[:2], pb[:2]):
        assert np.array_equal(u, v) if loss_func != "cosine_proximity" else _rel(u, np.asarray(v, np.float64)) < 5e-4


@pytest.mark.parametrize("dtype,strategy,B", [("f16x2", "batch_all", 150), ("bf16", "batch_hard", 150), ("bf16x3", "batch_all", 300)])
def test_gram_on_64x64_tiles_single_slab_equals_split_k_form(dtype, strategy, B):
    """The split 16-bit Gram matrix on 64 x 64 tiles over the whole K (gram64_kernel: ONE slab, the default) against the 128 x 128 split-K form whose
    slabs the miner's prologue sums (option gram64 = 0): the same products, summed in another order -- losses and gradients agree to fp32 rounding."""
    kw = dict(steps=2, seed=23, N=600, F=700, H=90, B=B)
    a, _, pa = _run_case(dtype, strategy, "cross_entropy", ("sigmoid", "sigmoid"), "gradient_descent", options={"gram64": 1}, **kw)
    b, _, pb = _run_case(dtype, strategy, "cross_entropy", ("sigmoid", "sigmoid"), "gradient_descent", options={"gram64": 0}, **kw)
    for (ra, sa, dWa, *_), (_, sb, dWb, *_) in zip(a, b):
        assert np.allclose(sa[:3], sb[:3], rtol=5e-6, atol=0), (sa, sb)
        if strategy == "batch_all":
            assert abs(sa[4] - sb[4]) <= 2 and sa[5] == sb[5]            # positive-triplet count (near-ties), N_valid
            assert abs(sa[2] - ra["triplet_loss"]) <= 1e-4 * abs(ra["triplet_loss"])
            assert _rel(dWa, np.asarray(dWb, np.float64)) < 2e-5


@pytest.mark.parametrize("dtype,strategy,shape", [("f16x2h", "batch_all", dict(N=600, F=700, H=90, B=150)), ("bf16x3", "batch_all", dict(N=700, F=900, H=500, B=300)),
                                                  ("f16x2h", "batch_hard", dict(N=600, F=700, H=200, B=260)), ("bf16", "batch_all", dict(N=600, F=700, H=130, B=64))])
def test_gram_three_products_in_one_stage_equals_the_k_concatenated_walk(dtype, strategy, shape):
    """gram64f_kernel (plan option gram_fused, default on): a stage holds the hi AND lo images of both row panels, the wave multiplies hi.hi, hi.lo, lo.hi from
    one set of fragments on three accumulators -- against gram64_kernel's walk over the K-concatenated operands [hi | hi | lo].[hi | lo | hi]^T (one
    accumulator): the same products, fp32 sums in another order."""
    from dae_rnn_news_recommendation_amd import _lib as L
    kw = dict(steps=2, seed=29, **shape)
    a, _, pa = _run_case(dtype, strategy, "cross_entropy", ("sigmoid", "sigmoid"), "gradient_descent", options={"gram_fused": 1}, **kw)
    try:
        b, _, pb = _run_case(dtype, strategy, "cross_entropy", ("sigmoid", "sigmoid"), "gradient_descent", options={"gram_fused": 0}, **kw)
    finally:
        L.set_glds_all(-20)
    for (ra, sa, dWa, *_), (_, sb, dWb, *_) in zip(a, b):
        assert np.allclose(sa[:3], sb[:3], rtol=5e-6, atol=0), (sa, sb)
        if strategy == "batch_all":
            assert abs(sa[4] - sb[4]) <= 2 and sa[5] == sb[5]            # positive-triplet count (near-ties), N_valid
            assert abs(sa[2] - ra["triplet_loss"]) <= (1e-4 if dtype != "bf16" else 5e-2) * abs(ra["triplet_loss"])
            assert _rel(dWa, np.asarray(dWb, np.float64)) < 2e-5


@pytest.mark.parametrize("dtype,dense,phase,opt,B", [("f16x2", False, 0, "gradient_descent", 150), ("f16x2", True, 0, "adam", 150), ("bf16", False, 1, "momentum", 200),
                                                     ("f16", False, 0, "ada_grad", 130), ("f16x2", False, 1, "gradient_descent", 64)])
def test_dw_transposed_a_form_equals_transposed_images(dtype, dense, phase, opt, B):
    """Plan option dw_tr (default: on for dense train sets, where it saves 38 us per step at F = 50000; off for CSR input, where it costs 3-5 us): the dW kernel consumes x~ and delta2 ROW-MAJOR through transposing LDS reads (gemm_dw_pc<TRA>) -- the decode stores
    delta2 once, the CSR scatter / dense gather write x~ instead of x~^T.  Against dw_tr = 0 (transposed operand images): the same MFMA products in the same K
    order, so gradients and updated parameters are bit-identical."""
    from dae_rnn_news_recommendation_amd import _lib as L
    L.load("f16" if dtype.startswith("f16") else "bf16").dae_set_glds(-5)          # the 160 x 128 kernel for every grid that fits one round (small test shapes)
    try:
        kw = dict(steps=3, seed=29, N=500, F=1000, H=200, B=B, dense=dense, phase=phase)
        a, ra, pa = _run_case(dtype, "batch_all", "cross_entropy" if not dense else "mean_squared", ("sigmoid", "sigmoid") if not dense else ("tanh", "none"),
                              opt, options={"dw_tr": 1}, **kw)
        b, rb, pb = _run_case(dtype, "batch_all", "cross_entropy" if not dense else "mean_squared", ("sigmoid", "sigmoid") if not dense else ("tanh", "none"),
                              opt, options={"dw_tr": 0}, **kw)
    finally:
        L.load("f16" if dtype.startswith("f16") else "bf16").dae_set_glds(-4)
    for (r, sa, dWa, dbha, dbva), (_, sb, dWb, dbhb, dbvb) in zip(a, b):
        assert np.array_equal(sa[:5], sb[:5]), (sa, sb)
        assert np.array_equal(dWa, dWb) and np.array_equal(dbha, dbhb) and np.array_equal(dbva, dbvb)
    assert _rel(a[0][2], a[0][0]["dW"]) < (2e-2 if dtype == "bf16" else 3e-3)            # ... and both at the oracle (first step: identical weights)
    for u, v in zip(pa, pb):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("dtype,loss_func,acts,strategy", [("f16x2", "cross_entropy", ("sigmoid", "sigmoid"), "batch_all"), ("f16x2", "mean_squared", ("tanh", "none"), "none"),
                                                          ("f16x2", "cosine_proximity", ("sigmoid", "sigmoid"), "batch_hard")])
def test_decode_paired_k_loop_equals_unpaired(dtype, loss_func, acts, strategy):
    """Plan option decode_pair: the decode's two W terms (h.W_hi + h.W_lo) as paired K-loop stages -- one h tile, both W tiles, (hi, lo) products interleaved per
    K tile -- against the walk over one segment after the other: the same products, fp32 sums in another order."""
    kw = dict(steps=2, seed=37, N=400, F=900, H=150, B=150)
    a, ra, pa = _run_case(dtype, strategy, loss_func, acts, "gradient_descent", options={"decode_pair": 1}, **kw)
    try:
        b, rb, pb = _run_case(dtype, strategy, loss_func, acts, "gradient_descent", options={"decode_pair": 0}, **kw)
    finally:
        pass
    for (r, sa, dWa, dbha, dbva), (_, sb, dWb, dbhb, dbvb) in zip(a, b):
        assert np.allclose(sa[:3], sb[:3], rtol=3e-6, atol=0), (sa, sb)
        assert _rel(dWa, np.asarray(dWb, np.float64)) < 1e-3                          # (a flipped fp16 rounding of delta2 = 2^-12 of one element)
        assert abs(sa[0] - r["cost"]) <= 5e-5 * abs(r["cost"])
    for u, v in zip(pa, pb):
        assert _rel(u, np.asarray(v, np.float64)) < 1e-3


@pytest.mark.parametrize("dtype,strategy,loss_func,acts,shape", [("f16x2h", "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), dict(N=400, F=900, H=150, B=150)),
                                                                ("f16x2h", "batch_hard", "cross_entropy", ("sigmoid", "sigmoid"), dict(N=500, F=1100, H=200, B=300)),
                                                                ("f16x2h", "none", "cross_entropy", ("sigmoid", "sigmoid"), dict(N=400, F=700, H=500, B=130)),
                                                                ("f16x2d", "none", "cross_entropy", ("sigmoid", "sigmoid"), dict(N=400, F=900, H=150, B=150)),
                                                                ("f16x2d", "batch_all", "cross_entropy", ("tanh", "sigmoid"), dict(N=400, F=700, H=500, B=130)),
                                                                ("f16x2", "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), dict(N=500, F=1100, H=200, B=300))])
def test_decode_register_carry_k_loops_equal_the_segment_walk(dtype, strategy, loss_func, acts, shape):
    """f16x2h keeps h AND W as hi + lo in the decode: z2 = h_hi.W_hi + h_hi.W_lo + h_lo.W_hi.  mainloop_n64_x3 (plan option decode_x3, default on, binary
    input) walks it in two stages per K tile -- (h_hi, W_hi), then (h_lo, W_lo) with the hi stage's fragments kept in registers -- instead of three K segments;
    f16x2d / f16x2 (W alone hi + lo: two segments over the same h) take mainloop_n64_c2, whose lo stage is the W_lo tile alone against the h fragments in
    registers (f16x2d: the kernel instantiation that also writes the lo images of delta2).  The same products, fp32 sums in another order.  Against the
    segment walk (decode_x3 = 0) and against the fp64 oracle."""
    from dae_rnn_news_recommendation_amd import _lib as L
    kw = dict(steps=2, seed=41, **shape)
    a, ra, pa = _run_case(dtype, strategy, loss_func, acts, "gradient_descent", options={"decode_x3": 1}, **kw)
    try:
        b, rb, pb = _run_case(dtype, strategy, loss_func, acts, "gradient_descent", options={"decode_x3": 0}, **kw)
    finally:
        L.set_glds_all(-18)
    for (r, sa, dWa, dbha, dbva), (_, sb, dWb, dbhb, dbvb) in zip(a, b):
        assert np.allclose(sa[:3], sb[:3], rtol=3e-6, atol=0), (sa, sb)
        assert _rel(dWa, np.asarray(dWb, np.float64)) < 1e-3 and _rel(dbha, np.asarray(dbhb, np.float64)) < 1e-3      # (a flipped fp16 rounding of delta2 = 2^-12 of one element)
        assert abs(sa[0] - r["cost"]) <= 3e-5 * abs(r["cost"]) and abs(sb[0] - r["cost"]) <= 3e-5 * abs(r["cost"]), (sa[0], sb[0], r["cost"])
        assert _rel(dWa, r["dW"]) < 2e-3, _rel(dWa, r["dW"])
    for u, v in zip(pa, pb):
        assert _rel(u, np.asarray(v, np.float64)) < 1e-3


@pytest.mark.parametrize("dtype", ["f16"])        # (the split modes always encode from the fp32 master: the library refuses encode_w32 = 0 there)
def test_step_f16_encode_from_the_16bit_shadow(dtype):
    """encode_w32 = 0 makes the sparse encode read the 16-bit shadow W_lo instead of the fp32 master (what dp.ShardedExchange selects for the non-split
    modes, where only the shadow is current on every rank).  In the fp16 build that shadow holds IEEE fp16 words: the kernel has to decode them as such
    (it used the bf16 bit trick there and h came out as garbage without an error).  Against the fp64 oracle, fp16-rounded weights."""
    out, ref, got = _run_case(dtype, "batch_all", "cross_entropy", ("sigmoid", "sigmoid"), "gradient_descent", steps=2, options={"encode_w32": 0})
    for r, st, dW, dbh, dbv in out:
        assert abs(st[0] - r["cost"]) <= 1e-3 * abs(r["cost"]), (st[0], r["cost"])          # h from 11-bit weights: ~2e-4 on the cost
        assert abs(st[2] - r["triplet_loss"]) <= 2e-3 * abs(r["triplet_loss"]), (st[2], r["triplet_loss"])
        assert _rel(dW, r["dW"]) < 2e-2 and _rel(dbh, r["dbh"]) < 2e-2, (_rel(dW, r["dW"]), _rel(dbh, r["dbh"]))


@pytest.mark.parametrize("dtype,strategy,loss_func,acts", [("f16x2", "batch_all", "cross_entropy", ("sigmoid", "sigmoid")), ("f16", "none", "mean_squared", ("tanh", "none")),
                                                           ("bf16", "batch_all", "cosine_proximity", ("sigmoid", "sigmoid"))])
def test_step_with_the_persistent_a_stationary_decode_kernel(dtype, strategy, loss_func, acts):
    """Plan option decode_ast = 1 (gemm_decode_ast: h fragments register-resident over the whole K, W streamed through a 6-slot LDS-DMA ring, persistent
    workgroups, wave-local epilogue, delta2^T straight from the accumulator layout) against the default tile kernel: the same products summed per K tile in
    the same order, so the losses agree to fp32 rounding of the partial sums and the gradients / parameters to the last bits of the 16-bit delta2 images.
    The kernel is OFF by default (measured 44 us against 38 at c2: profiles/r06_decode_ast.txt); this keeps its parity covered."""
    from dae_rnn_news_recommendation_amd import _lib as L
    kw = dict(steps=2, seed=23, N=500, F=1100, H=200, B=300)
    try:
        a, ra, pa = _run_case(dtype, strategy, loss_func, acts, "gradient_descent", options={"decode_ast": 1}, **kw)
    finally:
        L.set_glds_all(-15)
    b, rb, pb = _run_case(dtype, strategy, loss_func, acts, "gradient_descent", options={"decode_ast": 0}, **kw)
    for (r, sa, dWa, dbha, dbva), (_, sb, dWb, dbhb, dbvb) in zip(a, b):
        assert np.allclose(sa[:3], sb[:3], rtol=3e-6, atol=0), (sa, sb)
        assert abs(sa[0] - r["cost"]) <= (1e-4 if dtype != "bf16" else 2e-3) * abs(r["cost"])
        assert _rel(dWa, np.asarray(dWb, np.float64)) < 1e-3 and _rel(dbha, np.asarray(dbhb, np.float64)) < 1e-3, (_rel(dWa, np.asarray(dWb, np.float64)),)
        assert np.allclose(dbva, dbvb, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("dtype,strategy", [("f16x2h", "batch_all"), ("f16x2d", "none"), ("f16x2h", "batch_hard")])
def test_full_shape_steps_are_bit_identical_run_to_run(dtype, strategy):
    """The product defaults at the full c2 shape (F = 10000, H = 500, B = 800: 1106 decode tiles on the register-carry K loops, the fused-stage Gram, one
    round of the dW kernel): two engines fed the same rows, the same Philox corruption and the same W0 must end 12 steps with bit-identical parameters
    and statistics -- an LDS-DMA / barrier race in one of the hand-scheduled loops shows as a run-to-run difference at this occupancy, not at the small
    shapes of the oracle tests."""
    from dae_rnn_news_recommendation_amd import _lib as L
    from dae_rnn_news_recommendation_amd.engine import Engine
    from dae_rnn_news_recommendation_amd.synthetic import synthetic_csr, synthetic_labels, xavier_uniform
    N, F, H, B = 2400, 10000, 500, 800
    m = synthetic_csr(N, F, seed=5); lab = synthetic_labels(N, seed=5).astype(np.int32)
    W0 = xavier_uniform(F, H)
    outs = []
    for rep in range(2):
        eng = Engine(F, H, B, dtype=dtype, triplet=strategy, loss_func="cross_entropy", learning_rate=0.1)
        eng.upload_csr(m); eng.set_params(W0)
        stats = torch.zeros((12, L.STATS_STRIDE), device="cuda")
        for s in range(12):
            ids = (np.arange(B) + 800 * (s % 3)) % N
            ids = ids[np.argsort(lab[ids], kind="stable")]
            idx = torch.from_numpy(ids.astype(np.int32)).cuda()
            labs = torch.from_numpy(lab[ids]).cuda() if strategy != "none" else None
            eng.train_step(idx, labs, stats[s], corr_mode=L.CORR_PHILOX_MASK, seed=11, rng_stream=s, corr_frac=0.3, phase=3)
        torch.cuda.synchronize()
        outs.append(([np.asarray(x) for x in eng.get_params()], stats.cpu().numpy()))
    for u, v in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(u, v)
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.isfinite(outs[0][1][:, 0]).all() and outs[0][1][-1, 0] < outs[0][1][0, 0]          # ... and it trained
