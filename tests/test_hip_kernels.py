"""GPU parity tests, kernel by kernel: libdae_hip (through its C ABI) vs the CPU oracle / golden vectors.

Tolerances: integer / mask / count outputs bit-exact; fp32-mode floating point rtol 2e-5 (fp32
accumulation-order noise); bf16-mode operands are rounded to bf16 so element-wise checks use ~1e-2 and
loss-level checks 1e-4 relative (the gate BASELINE.json's north_star states)."""
import os

import numpy as np
import pytest
import torch
from scipy import sparse

import oracle as O

GATHER_TILE_DEFAULT = 0      # the library's default dense-gather tile (dae_gather.hip: g_gather_tile)

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


@pytest.fixture(scope="module")
def ops():
    from dae_rnn_news_recommendation_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def L():
    from dae_rnn_news_recommendation_amd import _lib
    _lib.load()
    return _lib


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def padded(a, rows, cols, dtype=torch.float32):
    t = torch.zeros((rows, cols), dtype=dtype, device="cuda")
    t[:a.shape[0], :a.shape[1]] = torch.as_tensor(np.asarray(a)).to(dtype).cuda()
    return t


def bf16_round(a):
    return torch.as_tensor(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


# ----------------------------------------------------------------------------------------------- #
# GEMM
# ----------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("glds", [1, 0])
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [(128, 128, 128, 0, 1), (256, 128, 512, 0, 1), (128, 384, 1024, 256, 3),
                                   (384, 256, 256, 128, 2), (896, 512, 2048, 0, 8)])
def test_gemm_nt(ops, L, glds, dtype, shape):
    M, N, K0, K1, splits = shape
    L.load().dae_set_glds(glds)
    try:
        rng = np.random.default_rng(M + N + K0 + K1)
        td = torch.bfloat16 if dtype == "bf16" else torch.float32
        A0 = torch.as_tensor(rng.standard_normal((M, K0)).astype(np.float32)).to(td).cuda()
        B0 = torch.as_tensor(rng.standard_normal((N, K0)).astype(np.float32)).to(td).cuda()   # asymmetric operands
        A1 = B1 = None
        ref = A0.double().cpu() @ B0.double().cpu().T
        if K1:
            A1 = torch.as_tensor(rng.standard_normal((M, K1)).astype(np.float32)).to(td).cuda()
            B1 = torch.as_tensor(rng.standard_normal((N, K1)).astype(np.float32)).to(td).cuda()
            ref = ref + A1.double().cpu() @ B1.double().cpu().T
        C = ops.gemm_nt(A0, B0, A1, B1, splits=splits)
        torch.cuda.synchronize()
        got = C.sum(0).double().cpu()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (shape, dtype, glds, err)
    finally:
        L.load().dae_set_glds(1)


@pytest.mark.parametrize("glds", [1, 0, -1])                       # LDS-DMA ring (8-wave where the grid fits), register staging, 4-wave DMA
@pytest.mark.parametrize("shape", [(256, 256, (128, 192, 64), 1), (896, 512, (512, 512, 512), 4), (896, 512, (10112, 10112, 10112, 896, 896), 8),
                                   (2048, 1536, (64, 128, 64, 128, 64), 1), (128, 128, (64, 0, 128), 2)])
def test_gemm_nt_segments(ops, L, glds, shape):
    """dae_gemm_nt_n: one contraction over 3..5 K segments with their own operands and leading dimensions (split-bf16 products);
    segment boundaries fall inside split-K slices, an empty segment is skipped."""
    M, N, Ks, splits = shape
    L.load().dae_set_glds(glds)
    try:
        rng = np.random.default_rng(M + N + sum(Ks))
        segs, ref = [], 0.0
        for i, K in enumerate(Ks):
            pad = 64 * (i % 3)                                                # different leading dimensions per segment
            a = torch.as_tensor(rng.standard_normal((M, K + pad)).astype(np.float32)).to(torch.bfloat16).cuda()[:, :K]
            b = torch.as_tensor(rng.standard_normal((N, K + 2 * pad)).astype(np.float32)).to(torch.bfloat16).cuda()[:, :K]
            segs.append((a, b))
            ref = ref + a.double().cpu() @ b.double().cpu().T
        C = ops.gemm_nt_n(segs, splits=splits)
        torch.cuda.synchronize()
        err = (C.sum(0).double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (shape, glds, err)
    finally:
        L.load().dae_set_glds(1)


def test_gemm_nt_split_bf16_products(ops, L):
    """x = hi + lo (both bf16): (hi,hi) + (hi,lo) + (lo,hi) in one contraction is a 2^-16-accurate product of the fp32 operands
    (plain bf16: 2^-8) -- the arithmetic of the Gram GEMM, here through the segment interface."""
    rng = np.random.default_rng(11)
    M, N, K = 256, 384, 1024
    a = torch.as_tensor(rng.standard_normal((M, K)).astype(np.float32)).cuda()
    b = torch.as_tensor(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    ah, bh = a.to(torch.bfloat16), b.to(torch.bfloat16)
    al, bl = (a - ah.float()).to(torch.bfloat16), (b - bh.float()).to(torch.bfloat16)
    ref = a.double().cpu() @ b.double().cpu().T
    scale = ref.abs().max().item()
    C3 = ops.gemm_nt_n([(ah, bh), (ah, bl), (al, bh)]).sum(0).double().cpu()
    C1 = ops.gemm_nt_n([(ah, bh)]).sum(0).double().cpu()
    e3, e1 = (C3 - ref).abs().max().item() / scale, (C1 - ref).abs().max().item() / scale
    assert e3 < 3e-5 and e1 > 20 * e3, (e1, e3)


@pytest.mark.parametrize("shape", [(896, 1024, 16384, 0), (1024, 1024, 16384, 0), (896, 1024, 16384, 896), (896, 768, 16384, 0)])
def test_gemm_nt_256_tile_kernel(ops, L, shape):
    """The 256 x 256 / 8-MFMA-wave kernel (dense-input encode / dh shapes): selected by its own split count, checked against
    float64 and against the 128 x 128 kernels on the same operands (partial last row tile: M = 896, 800)."""
    M, N, K0, K1 = shape
    lib = L.load()
    s = lib.dae_gemm_w8_splits(L.BF16, M, N, K0 + K1)
    assert s in (8, 16), s
    rng = np.random.default_rng(M + N + K0)
    A0 = torch.as_tensor(rng.standard_normal((M, K0)).astype(np.float32)).to(torch.bfloat16).cuda()
    B0 = torch.as_tensor(rng.standard_normal((N, K0)).astype(np.float32)).to(torch.bfloat16).cuda()
    A1 = B1 = None
    ref = A0.double().cpu() @ B0.double().cpu().T
    if K1:
        A1 = torch.as_tensor(rng.standard_normal((M, K1)).astype(np.float32)).to(torch.bfloat16).cuda()
        B1 = torch.as_tensor(rng.standard_normal((N, K1)).astype(np.float32)).to(torch.bfloat16).cuda()
        ref = ref + A1.double().cpu() @ B1.double().cpu().T
    for rep in range(3):                                 # repeated launches: a DMA / barrier race shows as a run-to-run difference
        C = ops.gemm_nt(A0, B0, A1, B1, splits=s)
        torch.cuda.synchronize()
        got = C.sum(0).double().cpu()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (shape, s, rep, err)
        if rep == 0:
            first = C.clone()
        else:
            assert torch.equal(C, first)
    if M % 128 == 0 and N % 128 == 0:
        lib.dae_set_glds(-6)
        try:
            C2 = ops.gemm_nt(A0, B0, A1, B1, splits=s)
            torch.cuda.synchronize()
        finally:
            lib.dae_set_glds(-7)
        e2 = (C2.sum(0) - first.sum(0)).abs().max().item() / ref.abs().max().item()
        assert e2 < 2e-6, e2


def test_gemm_identity_transpose_detecting(ops):
    """A = I picks out rows of Bt: catches a swapped C layout (guide 5.4 rule 16)."""
    M = N = K = 128
    A = torch.eye(M, K, dtype=torch.bfloat16, device="cuda")
    Bt = torch.as_tensor(np.arange(N * K, dtype=np.float32).reshape(N, K) % 251).to(torch.bfloat16).cuda()
    C = ops.gemm_nt(A, Bt)[0]
    assert torch.equal(C, Bt.float().T.contiguous())


# ----------------------------------------------------------------------------------------------- #
# gather
# ----------------------------------------------------------------------------------------------- #
def _rand_csr(rng, n, f, density, binary):
    m = sparse.random(n, f, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))),
                      format="csr", dtype=np.float32)
    m.data = np.ones_like(m.data) if binary else (m.data * 0.9 + 0.1).astype(np.float32)
    m.sort_indices()
    return m


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("binary", [True, False])
def test_gather_csr_keepbits(ops, L, dtype, binary):
    rng = np.random.default_rng(3)
    N, F, B = 300, 5000, 70          # F spans two 4096-column chunks; B not a multiple of anything
    m = _rand_csr(rng, N, F, 0.02, binary)
    m = sparse.vstack([m, sparse.csr_matrix((1, F), dtype=np.float32)]).tocsr()   # an empty row
    N += 1
    keep = rng.random(m.nnz) >= 0.3
    bits = np.packbits(keep, bitorder="little")
    bits = np.concatenate([bits, np.zeros((-len(bits)) % 4, np.uint8)]).view(np.uint32)
    rows = rng.permutation(N)[:B].astype(np.int32); rows[5] = N - 1
    dt = L.BF16 if dtype == "bf16" else L.F32
    x, xc, xct, rowsq = ops.gather_csr(dev(m.indptr.astype(np.int64)), dev(m.indices.astype(np.int32)),
                                       None if binary else dev(m.data), dev(rows), B, F, dt, want_rowsq=True,
                                       corr_mode=L.CORR_KEEPBITS, keep_bits=dev(bits.view(np.int32)), scale=0.5)
    torch.cuda.synchronize()
    dense = m.toarray()
    mc = m.copy(); mc.data = mc.data * keep * 0.5
    want_x = np.zeros((L.pad(B), L.pad(F)), np.float32); want_x[:B, :F] = dense[rows]
    want_xc = np.zeros_like(want_x); want_xc[:B, :F] = mc.toarray()[rows]
    if dtype == "bf16":
        want_x = bf16_round(want_x); want_xc = bf16_round(want_xc)
    assert np.array_equal(x.float().cpu().numpy(), want_x)
    assert np.array_equal(xc.float().cpu().numpy(), want_xc)
    assert np.array_equal(xct.float().cpu().numpy(), want_xc.T)
    assert np.allclose(rowsq.cpu().numpy()[:B], (dense[rows] ** 2).sum(1), rtol=1e-6)
    assert (rowsq.cpu().numpy()[B:] == 0).all()


def test_gather_csr_bits_and_encode_bits(ops, L):
    """Bit-packed x~ (binary CSR): the bit image equals packbits of the dense x~ image bit for bit, and the fused
    corrupt+encode GEMM on it reproduces the dense-operand GEMM (same bf16 products, fp32 accumulation)."""
    rng = np.random.default_rng(11)
    N, F, B, H = 500, 4500, 200, 250       # two 4096-column chunks, ragged everything
    m = _rand_csr(rng, N, F, 0.03, True)
    m = sparse.vstack([m, sparse.csr_matrix((1, F), dtype=np.float32)]).tocsr(); N += 1
    keep = rng.random(m.nnz) >= 0.3
    kb = np.packbits(keep, bitorder="little")
    kb = np.concatenate([kb, np.zeros((-len(kb)) % 4, np.uint8)]).view(np.int32)
    rows = rng.permutation(N)[:B].astype(np.int32); rows[7] = N - 1
    ip, ix, rw = dev(m.indptr.astype(np.int64)), dev(m.indices.astype(np.int32)), dev(rows)
    x, bits, xct = ops.gather_csr_bits(ip, ix, rw, B, F, L.BF16, corr_mode=L.CORR_KEEPBITS, keep_bits=dev(kb))
    x2, xc2, xct2, _ = ops.gather_csr(ip, ix, None, rw, B, F, L.BF16, corr_mode=L.CORR_KEEPBITS, keep_bits=dev(kb))
    torch.cuda.synchronize()
    Bp, Fp, Hp = L.pad(B), L.pad(F), L.pad(H)
    want = np.packbits(xc2.float().cpu().numpy() != 0, axis=1, bitorder="little").view(np.uint32)
    assert want.shape == (Bp, Fp // 32)
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), want)
    assert torch.equal(x, x2) and torch.equal(xct, xct2)
    Wt = padded(rng.standard_normal((H, F)).astype(np.float32) * 0.1, Hp, Fp, torch.bfloat16)
    ref = xc2.double() @ Wt.double().T
    for splits in (1, 3, 8):
        z = ops.encode_bits(bits, Wt, splits).sum(0)
        zd = ops.gemm_nt(xc2, Wt, splits=splits).sum(0)
        torch.cuda.synchronize()
        assert rel_err(z.cpu().numpy(), ref.cpu().numpy()) < 2e-6
        assert rel_err(z.cpu().numpy(), zd.cpu().numpy()) < 2e-6
        assert (z[B:] == 0).all()


def test_salt_pepper_batch_matches_oracle(L):
    """Device salt-and-pepper (dae_salt_pepper_batch) == the oracle's restatement of utils.salt_and_pepper_noise with the
    Philox stream: same flipped columns, later draws win, sorted batch-local CSR, zeros dropped."""
    from dae_rnn_news_recommendation_amd.engine import Engine
    rng = np.random.default_rng(8)
    N, F, B, v = 90, 1200, 50, 300
    for binary in (True, False):
        m = _rand_csr(rng, N, F, 0.04, binary)
        eng = Engine(F, 64, B, dtype="fp32")
        eng.upload_csr(m)
        rows = rng.permutation(N)[:B].astype(np.int32)
        lo, hi = 0.0, float(m.data.max())
        c = eng.salt_pepper_batch(dev(rows), v, lo, hi, seed=0x1234ABCD5678, rng_stream=3)
        torch.cuda.synchronize()
        span = c["indptr"].cpu().numpy(); ci = c["indices"].cpu().numpy(); cv = c["values"].cpu().numpy()
        want = O.salt_and_pepper_philox(m, rows, v, 0x1234ABCD5678, 3, lo=lo, hi=hi)
        got = np.zeros((B, F))
        for i in range(B):
            s0, e0 = span[2 * i], span[2 * i + 1]
            cols = ci[s0:e0]
            assert (np.diff(cols) > 0).all() and (cv[s0:e0] != 0).all()
            got[i, cols] = cv[s0:e0]
        assert np.array_equal(got.astype(np.float32), want.astype(np.float32))
        assert (got != m[rows].toarray()).any()


def test_gather_csr_philox_matches_oracle(ops, L):
    rng = np.random.default_rng(4)
    N, F, B = 200, 1000, 128
    m = _rand_csr(rng, N, F, 0.05, True)
    rows = rng.permutation(N)[:B].astype(np.int32)
    seed, stream, frac = 0x1234567890ABCDEF, 7, 0.3
    _, xc, _, _ = ops.gather_csr(dev(m.indptr.astype(np.int64)), dev(m.indices.astype(np.int32)), None, dev(rows), B, F,
                                 L.F32, corr_mode=L.CORR_PHILOX_MASK, seed=seed, rng_stream=stream, corr_frac=frac)
    keep = O.philox_uniform(np.arange(m.nnz, dtype=np.uint64), seed, stream) >= np.float32(frac)
    mc = m.copy(); mc.data = mc.data * keep
    want = np.zeros((L.pad(B), L.pad(F)), np.float32); want[:B, :F] = mc.toarray()[rows]
    assert np.array_equal(xc.cpu().numpy(), want)
    assert abs(keep.mean() - 0.7) < 0.02


@pytest.mark.parametrize("tile", [0, 1, 2, 3])     # tile shape of the kernel: 64 x 64, 64 x 128, 128 x 64, 128 x 128 (rows x features)
@pytest.mark.parametrize("F", [300, 301])          # 301: rows not 16-byte aligned -> the element-wise variant of the kernel
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_gather_dense(ops, L, dtype, F, tile):
    from dae_rnn_news_recommendation_amd.engine import Engine
    rng = np.random.default_rng(5)
    N, B = 150, 70 if tile < 2 else 200              # 200 rows: two row tiles of 128, the second one ragged
    N = max(N, B + 20)
    data = (rng.random((N, F)) * (rng.random((N, F)) < 0.3)).astype(np.float32)
    rows = rng.permutation(N)[:B].astype(np.int32)
    seed, stream, frac = 99, 3, 0.25
    dt = L.BF16 if dtype == "bf16" else L.F32
    eng = Engine(8, 4, 8)                              # (the option is process-wide; any plan can set it)
    try:
        eng.set_option("gather_tile", tile)
        x, xc, xct, rowsq = ops.gather_dense(dev(data), dev(rows), B, F, dt, want_rowsq=True, corr_mode=L.CORR_PHILOX_MASK,
                                             seed=seed, rng_stream=stream, corr_frac=frac, scale=1.0)
    finally:
        eng.set_option("gather_tile", GATHER_TILE_DEFAULT)
    keep = O.philox_uniform_dense(rows, F, seed, stream) >= np.float32(frac)
    want_x = np.zeros((L.pad(B), L.pad(F)), np.float32); want_x[:B, :F] = data[rows]
    want_xc = np.zeros_like(want_x); want_xc[:B, :F] = data[rows] * keep
    if dtype == "bf16":
        want_x = bf16_round(want_x); want_xc = bf16_round(want_xc)
    assert np.array_equal(x.float().cpu().numpy(), want_x)
    assert np.array_equal(xc.float().cpu().numpy(), want_xc)
    assert np.array_equal(xct.float().cpu().numpy(), want_xc.T)
    assert np.allclose(rowsq.cpu().numpy()[:B], (data[rows] ** 2).sum(1), rtol=1e-5)


# ----------------------------------------------------------------------------------------------- #
# encode / decode epilogues
# ----------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("act", ["sigmoid", "tanh", "none"])
def test_encode_finish(ops, L, act):
    rng = np.random.default_rng(6)
    B, H, S = 70, 200, 3
    Bp, Hp = L.pad(B), L.pad(H)
    slabs = rng.standard_normal((S, Bp, Hp)).astype(np.float32)
    bh = np.zeros(Hp, np.float32); bh[:H] = rng.standard_normal(H) * 0.3
    h32, hlo, ht, ha, hb = ops.encode_finish(dev(slabs), dev(bh), B, H, L.ACT[act], L.BF16, want_hcat=True)
    z = slabs.sum(0)[:B, :H] + bh[:H]
    want = np.zeros((Bp, Hp), np.float32); want[:B, :H] = O.act(act, z) - O.act(act, bh[:H])
    assert np.allclose(h32.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
    assert np.array_equal(hlo.float().cpu().numpy(), bf16_round(h32.cpu().numpy()))
    assert np.array_equal(ht.float().cpu().numpy(), hlo.float().cpu().numpy().T)
    # split-bf16 Gram operands: hcat_a . hcat_b^T == h h^T to ~2^-16 of |h|^2 (vs 2^-8 for plain bf16)
    D = ops.gemm_nt(ha, hb)[0].cpu().numpy()
    hd = h32.double().cpu().numpy()
    ref = hd @ hd.T
    assert np.abs(D - ref).max() <= 3e-5 * np.abs(ref).max()
    assert np.array_equal(ha[:, :Hp].float().cpu().numpy(), hlo.float().cpu().numpy())


def test_label_stats_arbitrary_ids(ops, L):
    """ids outside the LDS histogram range take the comparison path; results are the same integers."""
    rng = np.random.default_rng(12)
    B = 300
    small = rng.integers(0, 5, B)
    big = (small * 1000003 + 77777).astype(np.int32)
    neg = (small - 3).astype(np.int32)
    outs = []
    for lab in (small.astype(np.int32), big, neg):
        nvalid, dw, cw = ops.label_stats(dev(lab), B, L.TRIPLET["batch_all"])
        outs.append((int(nvalid.item()), dw.cpu().numpy()[:B].copy(), cw.cpu().numpy().copy()))
    nv, dwc = O.batch_all_closed_form(small)
    for o in outs:
        assert o[0] == nv and np.array_equal(o[1], dwc) and np.array_equal(o[2], outs[0][2])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("loss_func,dec_act", [("cross_entropy", "sigmoid"), ("mean_squared", "none"),
                                               ("mean_squared", "tanh"), ("cosine_proximity", "sigmoid")])
def test_decode_loss(ops, L, dtype, loss_func, dec_act):
    rng = np.random.default_rng(7)
    B, F, H = 150, 300, 100
    Bp, Fp, Hp = L.pad(B), L.pad(F), L.pad(H)
    dt = L.BF16 if dtype == "bf16" else L.F32
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    h = (rng.random((B, H)) - 0.5).astype(np.float32)
    W = rng.uniform(-0.2, 0.2, (F, H)).astype(np.float32)
    bv = (rng.standard_normal(F) * 0.1).astype(np.float32)
    x = (rng.random((B, F)) < 0.1).astype(np.float32)
    if loss_func != "cross_entropy":
        x = x * rng.random((B, F)).astype(np.float32)
    w = rng.integers(0, 50, B).astype(np.float32)
    cw = np.zeros(Bp, np.float32); cw[:B] = w / (w.sum() + 1e-16)
    if dtype == "bf16":
        h, W, x = bf16_round(h), bf16_round(W), bf16_round(x)
    bvp = np.zeros(Fp, np.float32); bvp[:F] = bv
    args = (padded(h, Bp, Hp, td), padded(W, Fp, Hp, td), dev(bvp), padded(x, Bp, Fp, td), dev(cw), B, F, H,
            L.ACT[dec_act], L.LOSS[loss_func], dt)
    # oracle
    y = O.decode(h, W, bv, dec_act, np.float64)
    rows = O.weighted_loss_rows(x.astype(np.float64), y, loss_func, np.float64)
    dy = cw[:B, None].astype(np.float64) * O.dae_oracle._loss_dy(x.astype(np.float64), y, loss_func, np.float64)
    d2 = dy * O.act_grad_from_output(dec_act, y)
    if loss_func == "cosine_proximity":
        cos_stats = torch.zeros(3 * Bp, device="cuda")
        cos_stats[:B] = dev((x.astype(np.float64) ** 2).sum(1).astype(np.float32))
        r1 = ops.decode_loss(*args, cos_pass=1, cos_stats=cos_stats)
        rowloss = ops.cos_reduce(r1["cos_part"], B, cos_stats)
        r = ops.decode_loss(*args, cos_pass=2, cos_stats=cos_stats)
        got_rows = rowloss.cpu().numpy()[:B]
    else:
        r = ops.decode_loss(*args)
        got_rows = r["rowloss_part"].sum(0).cpu().numpy()[:B]
        assert (r["rowloss_part"].sum(0).cpu().numpy()[B:] == 0).all()
        # per-tile share of sum_i cw_i * rowloss_i (what the step driver consumes)
        assert abs(r["tile_part"].double().sum().item() - float((cw[:B].astype(np.float64) * rows).sum())) <= 2e-5 * abs(float((cw[:B] * rows).sum()))
    tol = 2e-5 if dtype == "f32" else 2e-5      # operands are pre-rounded: only accumulation order differs
    assert rel_err(got_rows, rows) < tol
    got_d2 = r["delta2"].float().cpu().numpy()
    d2tol = 1e-5 if dtype == "f32" else 6e-3          # delta2 itself is stored in bf16
    assert rel_err(got_d2[:B, :F], d2) < d2tol
    assert (got_d2[B:] == 0).all() and (got_d2[:, F:] == 0).all()
    assert np.array_equal(r["delta2_t"].float().cpu().numpy(), got_d2.T)
    assert rel_err(r["dbv_part"].sum(0).cpu().numpy()[:F], d2.sum(0)) < 2e-5


def test_decode_ce_saturation(ops, L):
    """y saturates to exactly 0/1 in fp32: log(1e-16) terms and 0*1e16 gradients stay finite."""
    B, F, H = 128, 128, 128
    h = np.zeros((B, H), np.float32); h[:, 0] = 1.0
    W = np.zeros((F, H), np.float32); W[:64, 0] = 60.0; W[64:, 0] = -120.0
    x = np.zeros((B, F), np.float32); x[:, ::2] = 1.0
    cw = np.full(B, 1.0 / B, np.float32)
    r = ops.decode_loss(dev(h), dev(W), dev(np.zeros(F, np.float32)), dev(x), dev(cw), B, F, H, L.ACT["sigmoid"],
                        L.LOSS["cross_entropy"], L.F32)
    rows = r["rowloss_part"].sum(0).cpu().numpy()
    d2 = r["delta2"].cpu().numpy()
    assert np.isfinite(rows).all() and np.isfinite(d2).all()
    y = O.decode(h, W, np.zeros(F, np.float32), "sigmoid", np.float32)
    want = O.weighted_loss_rows(x, y, "cross_entropy", np.float32)
    assert np.allclose(rows, want, rtol=1e-4)


# ----------------------------------------------------------------------------------------------- #
# miners
# ----------------------------------------------------------------------------------------------- #
def _gram_slabs(ops, L, h, splits):
    B, H = h.shape
    hp = padded(h, L.pad(B), L.pad(H))
    return ops.gram(hp, splits)


@pytest.mark.parametrize("case", range(int(G["n_miner_cases"])))
def test_miners_vs_reference_golden(ops, L, case):
    """Golden vectors produced by the reference's own triplet_loss_utils.py (tests/golden/make_golden.py)."""
    k = f"miner{case}_"
    lab, h = G[k + "labels"], G[k + "encode"]
    B = len(lab)
    labels = dev(lab.astype(np.int32))
    D = _gram_slabs(ops, L, h, 2)
    assert np.allclose(D.sum(0).cpu().numpy()[:B, :B], h.astype(np.float64) @ h.astype(np.float64).T, rtol=1e-5, atol=1e-6)
    nvalid, dw, cw = ops.label_stats(labels, B, L.TRIPLET["batch_all"])
    assert np.array_equal(dw.cpu().numpy()[:B], G[k + "ba_all_dw"].astype(np.int64))        # bit exact
    assert int(nvalid.item()) == int(G[k + "mask3"].sum())
    for pos_only, s in ((False, "all"), (True, "pos")):
        lp, npos, Gm, role = ops.triplet_batch_all(D, labels, B, pos_only)
        cw2 = cw.clone()
        tri, dwf = ops.triplet_finalize(L.TRIPLET["batch_all"], pos_only, B, 1.0, lp, npos, nvalid, None, role, cw2)
        tri = tri.cpu().numpy()
        assert np.allclose(tri[1], G[k + f"ba_{s}_loss"], rtol=1e-5, atol=1e-7)
        assert np.allclose(tri[2], G[k + f"ba_{s}_frac"], rtol=1e-6)
        assert tri[3] == G[k + f"ba_{s}_num"]
        if pos_only:
            assert np.array_equal(dwf.cpu().numpy()[:B], G[k + "ba_pos_dw"])
    lp, cnt, dwi, Gm = ops.triplet_batch_hard(D, labels, B)
    cw3 = torch.zeros_like(cw)
    tri, dwf = ops.triplet_finalize(L.TRIPLET["batch_hard"], False, B, 1.0, lp, cnt, nvalid, dwi, None, cw3)
    tri = tri.cpu().numpy()
    # the golden D was computed by NumPy's matmul; equality-based data_weight needs OUR D to be self-consistent,
    # so compare against the oracle evaluated on the kernel's own Gram matrix.
    Dk = D.sum(0).cpu().numpy()[:B, :B]
    lo, dwo, fro, numo = O.batch_hard_triplet_loss(lab, h, np.float32, D=Dk)
    assert np.allclose(tri[1], lo, rtol=1e-5, atol=1e-7) and tri[3] == numo and np.allclose(tri[2], fro)
    assert np.array_equal(dwf.cpu().numpy()[:B], dwo)
    assert np.allclose(tri[1], G[k + "bh_loss"], rtol=1e-4, atol=1e-6)      # and close to the reference's own value


@pytest.mark.parametrize("B,classes,signed,scale", [(200, 4, True, 2.0), (333, 7, True, 2.0), (128, 1, False, 0.3), (257, 50, True, 2.0),
                                                     (300, 3, True, 12.0),      # D row range > 80: the direct (non-factorised) sweep
                                                     (300, 3, True, 1.5), (300, 3, True, 2.8),   # row ranges ~7 / ~25: 8 / 2 factors per logarithm
                                                     (300, 3, True, 4.0),       # row range in (40, 80]: per-cell sweep (exact), scaled pairs (fast)
                                                     (800, 4, True, 2.2),       # the c2 batch size: two chunks of negatives
                                                     (1100, 5, True, 1.0)])     # B > 1024: two-kernel label statistics
def test_miners_gradients(ops, L, B, classes, signed, scale):
    rng = np.random.default_rng(B)
    H = 40
    h = (rng.random((B, H)).astype(np.float32) - (0.5 if signed else 0.0)) * scale
    lab = rng.integers(0, classes, B)
    labels = dev(lab.astype(np.int32))
    D = _gram_slabs(ops, L, h, 3)
    Dk = D.sum(0).cpu().numpy()[:B, :B]
    nvalid, dw, cw = ops.label_stats(labels, B, L.TRIPLET["batch_all"])
    nv, dwc = O.batch_all_closed_form(lab)
    assert int(nvalid.item()) == nv and np.array_equal(dw.cpu().numpy()[:B], dwc)
    if nv > 0:
        assert np.allclose(cw.cpu().numpy()[:B], dwc / (dwc.sum() + 1e-16), rtol=1e-6)
    # batch_all
    lp, npos, Gm, _ = ops.triplet_batch_all(D, labels, B, False)
    tri, _ = ops.triplet_finalize(L.TRIPLET["batch_all"], False, B, 0.5, lp, npos, nvalid, None, None, cw)
    lo, dwo, fro, numo, Go = O.batch_all_triplet_loss(lab, h, False, np.float64, return_grad=True, D=Dk.astype(np.float64))
    tri = tri.cpu().numpy()
    assert np.allclose(tri[1], lo, rtol=2e-5, atol=1e-7)
    lo32, _, _, num32 = O.batch_all_triplet_loss(lab, h, False, np.float32, D=Dk)
    assert tri[3] == num32                                      # positive-triplet count: bit exact on the same D
    Gs = ops.sym_scale(Gm, B, dev(tri), L.F32).cpu().numpy()
    want = 0.5 * (Go + Go.T)
    assert rel_err(Gs[:B, :B], want) < 5e-5 if nv else (Gs == 0).all()
    assert (Gs[B:] == 0).all() and (Gs[:, B:] == 0).all()
    # the bf16 steps' FAST mode (sums of 1/w, no first-order log1p correction, exact re-run of anchors it cannot bound)
    lpf, nposf, Gf, _ = ops.triplet_batch_all(D, labels, B, False, fast=True)
    trif, _ = ops.triplet_finalize(L.TRIPLET["batch_all"], False, B, 0.5, lpf, nposf, nvalid, None, None, cw)
    trif = trif.cpu().numpy()
    assert np.allclose(trif[1], lo, rtol=2e-5, atol=1e-7) and trif[3] == num32
    Gsf = ops.sym_scale(Gf, B, dev(trif), L.F32).cpu().numpy()
    assert rel_err(Gsf[:B, :B], want) < 1e-4 if nv else (Gsf == 0).all()
    # batch_hard
    lp, cnt, dwi, Gm = ops.triplet_batch_hard(D, labels, B)
    cwh = torch.zeros_like(cw)
    tri, dwf = ops.triplet_finalize(L.TRIPLET["batch_hard"], False, B, 0.5, lp, cnt, nvalid, dwi, None, cwh)
    lo, dwo, fro, numo, Go = O.batch_hard_triplet_loss(lab, h, np.float32, return_grad=True, D=Dk)
    tri = tri.cpu().numpy()
    assert np.allclose(tri[1], lo, rtol=1e-5, atol=1e-7) and tri[3] == numo
    assert np.array_equal(dwf.cpu().numpy()[:B], dwo)
    if dwo.sum() > 0:
        assert np.allclose(cwh.cpu().numpy()[:B], dwo / (dwo.sum() + 1e-16), rtol=1e-6)
    Gs = ops.sym_scale(Gm, B, dev(tri), L.F32).cpu().numpy()
    want = 0.5 * (Go.astype(np.float64) + Go.T)
    assert rel_err(Gs[:B, :B], want) < 5e-5 if numo else (Gs == 0).all()


def test_label_stats_none(ops, L):
    nvalid, dw, cw = ops.label_stats(None, 70, L.TRIPLET["none"])
    c = cw.cpu().numpy()
    assert np.allclose(c[:70], 1.0 / 70) and (c[70:] == 0).all()


# ----------------------------------------------------------------------------------------------- #
# backward pieces + optimizer
# ----------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("act", ["sigmoid", "tanh", "none"])
def test_dh_finish_and_bias_grads(ops, L, act):
    rng = np.random.default_rng(8)
    B, H, F, S = 150, 100, 300, 2
    Bp, Hp, Fp = L.pad(B), L.pad(H), L.pad(F)
    slabs = rng.standard_normal((S, Bp, Hp)).astype(np.float32)
    bh = np.zeros(Hp, np.float32); bh[:H] = rng.standard_normal(H) * 0.3
    z1 = rng.standard_normal((B, H)).astype(np.float32)
    a1 = O.act(act, z1)
    h = np.zeros((Bp, Hp), np.float32); h[:B, :H] = a1 - O.act(act, bh[:H])
    d1t, colsum, d1 = ops.dh_finish(dev(slabs), dev(h), dev(bh), B, H, L.ACT[act], L.F32)
    dh = slabs.sum(0)[:B, :H].astype(np.float64)
    want_d1 = dh * O.act_grad_from_output(act, (h[:B, :H] + O.act(act, bh[:H])).astype(np.float64))
    got = d1.cpu().numpy()
    assert rel_err(got[:B, :H], want_d1) < 1e-5 and (got[B:] == 0).all() and (got[:, H:] == 0).all()
    assert np.array_equal(d1t.cpu().numpy(), got.T)
    dbv_part = rng.standard_normal((2 * Bp // 128, Fp)).astype(np.float32)
    dbh, dbv = ops.bias_grads(dev(dbv_part), colsum, dev(bh), H, F, L.ACT[act])
    ab = O.act(act, bh[:H].astype(np.float64))
    want_dbh = want_d1.sum(0) - O.act_grad_from_output(act, ab) * dh.sum(0)
    assert rel_err(dbh.cpu().numpy()[:H], want_dbh) < 2e-5 and (dbh.cpu().numpy()[H:] == 0).all()
    assert np.allclose(dbv.cpu().numpy()[:F], dbv_part.sum(0)[:F], rtol=1e-5, atol=1e-6)
    assert (dbv.cpu().numpy()[F:] == 0).all()


@pytest.mark.parametrize("opt", ["gradient_descent", "ada_grad", "momentum", "adam"])
def test_opt_step(ops, L, opt):
    rng = np.random.default_rng(9)
    Fp, Hp = 256, 128
    W = rng.standard_normal((Fp, Hp)).astype(np.float32); bh = rng.standard_normal(Hp).astype(np.float32)
    bv = rng.standard_normal(Fp).astype(np.float32)
    n = Fp * Hp + Hp + Fp
    dW, dbh, dbv = W.copy(), bh.copy(), bv.copy()
    st = O.OptState(opt, [W.shape, bh.shape, bv.shape], np.float32)
    tW, tbh, tbv = dev(W), dev(bh), dev(bv)
    s1 = None if opt == "gradient_descent" else (torch.full((n,), 0.1, device="cuda") if opt == "ada_grad" else torch.zeros(n, device="cuda"))
    s2 = torch.zeros(n, device="cuda") if opt == "adam" else None
    Wlo = torch.zeros((Fp, Hp), dtype=torch.bfloat16, device="cuda"); Wtlo = torch.zeros((Hp, Fp), dtype=torch.bfloat16, device="cuda")
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        lr = 0.1
        lr_dev = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) if opt == "adam" else lr
        ops.opt_step(L.OPT[opt], lr_dev, 0.5, 0.5, tW, tbh, tbv, dev(g), s1, s2, L.BF16, Wlo, Wtlo)
        gs = g * np.float32(0.5)
        O.opt_apply(st, [dW, dbh, dbv], [gs[:Fp * Hp].reshape(Fp, Hp), gs[Fp * Hp:Fp * Hp + Hp], gs[Fp * Hp + Hp:]], lr, 0.5, np.float32)
    assert np.allclose(tW.cpu().numpy(), dW, rtol=2e-5, atol=2e-6)
    assert np.allclose(tbh.cpu().numpy(), dbh, rtol=2e-5, atol=2e-6)
    assert np.allclose(tbv.cpu().numpy(), dbv, rtol=2e-5, atol=2e-6)
    assert np.array_equal(Wlo.float().cpu().numpy(), bf16_round(tW.cpu().numpy()))
    assert np.array_equal(Wtlo.float().cpu().numpy(), bf16_round(tW.cpu().numpy()).T)


def test_explicit_triplet(ops, L):
    rng = np.random.default_rng(10)
    B, H = 50, 70
    Hp = L.pad(H)
    h3 = np.zeros((L.pad(3 * B), Hp), np.float32)
    h3[:3 * B, :H] = rng.random((3 * B, H)) - 0.5
    dh3, lp, tri = ops.explicit_triplet(dev(h3), B, H, 2.0)
    ho, hp_, hn = h3[:B, :H].astype(np.float64), h3[B:2 * B, :H].astype(np.float64), h3[2 * B:3 * B, :H].astype(np.float64)
    t = (ho * hn - ho * hp_).sum(1)
    assert np.allclose(tri.cpu().numpy()[1], np.mean(np.logaddexp(0, t)), rtol=1e-5)
    g = 2.0 * O.sigmoid(t)[:, None] / B
    got = dh3.cpu().numpy()
    assert rel_err(got[:B, :H], g * (hn - hp_)) < 1e-5
    assert rel_err(got[B:2 * B, :H], -g * ho) < 1e-5 and rel_err(got[2 * B:3 * B, :H], g * ho) < 1e-5
