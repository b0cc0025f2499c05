"""csrc/gemm_x3.hip (split-bf16 GEMM, eegclip_gemm_desc.precision = BF16X3) through the C ABI on both backends: every tile
configuration x operand layout, ragged edges, k tails, split-K, row sums and the shared epilogue.  Two references: the kernel's own
arithmetic restated in numpy (a_hi b_hi + a_hi b_lo + a_lo b_hi over bf16-rounded halves, float64 sums) -- tight, proves the three
products and the split are what the header says -- and the exact fp64 product, with the 2^-16 |a||b| per-term bound."""
import ctypes

import numpy as np
import pytest

from backends import be  # noqa: F401
from eeg_image_decode_amd import _abi
from philox_np import keep_mask
from test_kernels_gemm import f32, mk, run

D = _abi.dim
CFGS = [0, 1, 2, 3, 4, 5]        # 64x64x32, +double buffer, 64x64x64, +db, 128x128x32, +db


def prec(cfg=None):
    return _abi.PREC_BF16X3 | ((cfg + 1) << 8 if cfg is not None else 0)


def bf16_round(x):
    """float32 array -> nearest-even bf16, returned as float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_round(x)
    lo = bf16_round((x - hi).astype(np.float32))
    return hi.astype(np.float64), lo.astype(np.float64)


def x3_ref(a, b):
    """(M,K) @ (K,N) with the kernel's three products"""
    ah, al = split(a)
    bh, bl = split(b)
    return ah @ bh + ah @ bl + al @ bh


@pytest.mark.parametrize("cfg", CFGS)
@pytest.mark.parametrize("M,N,K,split_k", [(2, 2, 2, 1), (130, 70, 50, 1), (64, 128, 96, 1), (250, 744, 250, 1), (72, 66, 330, 3), (200, 130, 70, 2)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_x3_layouts_tiles_split_k(be, cfg, M, N, K, split_k, ta, tb):
    if be.name == "emu" and cfg in (2, 3) and (M, N, K) != (130, 70, 50):
        pytest.skip("emulator time: BK = 64 shares every code path with BK = 32; one shape suffices on the CPU")
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    a = f32(rng, *((K, M) if ta else (M, K)))
    b = f32(rng, *((N, K) if tb else (K, N)))
    bias, c0 = f32(rng, N), f32(rng, M, N)
    A, B, BI, C = be.dev(a), be.dev(b), be.dev(bias), be.dev(c0 if split_k > 1 else np.full((M, N), np.nan, np.float32))
    Am, Ak = (D(1), D(M)) if ta else (D(K), D(1))
    Bk, Bn = (D(1), D(K)) if tb else (D(N), D(1))
    run(be, mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), bias_n=be.ptr(BI), split_k=split_k, precision=prec(cfg)))
    am, bm = (a.T if ta else a), (b.T if tb else b)
    extra = bias + (c0 if split_k > 1 else 0)
    got = be.host(C)
    np.testing.assert_allclose(got, x3_ref(am, bm) + extra, atol=4e-6 * max(1.0, np.abs(got).max()))        # the kernel's own arithmetic
    exact = am.astype(np.float64) @ bm.astype(np.float64) + extra
    bound = 2.0 ** -16 * (np.abs(am).astype(np.float64) @ np.abs(bm).astype(np.float64)) + 1e-5              # dropped lo*lo + rounding of lo
    assert (np.abs(got - exact) <= bound).all()


@pytest.mark.parametrize("cfg", [1, 5])
def test_x3_is_much_closer_to_fp32_than_one_bf16_product(be, cfg):
    rng = np.random.default_rng(3)
    M, N, K = 96, 80, 250
    a, w = f32(rng, M, K), f32(rng, N, K)
    A, W, C = be.dev(a), be.dev(w), be.zeros((M, N))
    run(be, mk(be, M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1), precision=prec(cfg)))
    exact = a.astype(np.float64) @ w.T.astype(np.float64)
    one = bf16_round(a).astype(np.float64) @ bf16_round(w).T.astype(np.float64)
    e3, e1 = np.abs(be.host(C) - exact).max(), np.abs(one - exact).max()
    assert e3 < 3e-4 and e3 < e1 / 100, (e3, e1)


@pytest.mark.parametrize("cfg", [0, 1, 5])
@pytest.mark.parametrize("M,N,K,split_k", [(130, 70, 50, 1), (72, 66, 330, 3), (200, 40, 98, 2)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_x3_rowsum_a(be, cfg, M, N, K, split_k, ta, tb):
    rng = np.random.default_rng(M + N + K + ta * 2 + tb)
    a = f32(rng, *((K, M) if ta else (M, K)))
    b = f32(rng, *((N, K) if tb else (K, N)))
    r0 = f32(rng, M)
    A, B, C, RS = be.dev(a), be.dev(b), be.zeros((M, N)), be.dev(r0)
    Am, Ak = (D(1), D(M)) if ta else (D(K), D(1))
    Bk, Bn = (D(1), D(K)) if tb else (D(N), D(1))
    run(be, mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), split_k=split_k, accumulate=int(split_k > 1), rowsum_a=be.ptr(RS), precision=prec(cfg)))
    am, bm = (a.T if ta else a), (b.T if tb else b)
    np.testing.assert_allclose(be.host(C), x3_ref(am, bm), atol=1e-4)
    np.testing.assert_allclose(be.host(RS), r0 + am.astype(np.float64).sum(1), atol=2e-4)


@pytest.mark.parametrize("N", [92, 90])                         # 90: rows are not a multiple of 4 words -> per-element Philox, ragged last tile
@pytest.mark.parametrize("cfg", [1, 5])
def test_x3_full_epilogue(be, cfg, N):
    """bias -> Cpre -> GELU -> dropout (Philox of the logical index) -> residual -> accumulate; then the GELU' form; then the two-level C map
    of the value embedding (rows 1..63 of every 64-row token block + PE)"""
    from scipy.special import erf
    rng = np.random.default_rng(5)
    M, K = 70, 40
    a, w, bn, bm, r, c0 = f32(rng, M, K), f32(rng, N, K), f32(rng, N), f32(rng, M), f32(rng, M, N), f32(rng, M, N)
    A, W, BN, BM, R, C, CP = be.dev(a), be.dev(w), be.dev(bn), be.dev(bm), be.dev(r), be.dev(c0), be.zeros((M, N))
    p, seed, site = 0.25, 0x1234567890ABCDEF, 3
    run(be, mk(be, M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1), Cpre=be.ptr(CP), bias_n=be.ptr(BN), bias_m=be.ptr(BM),
               R=be.ptr(R), Rm=D(N), Rn=D(1), act=_abi.ACT_GELU, accumulate=1, drop_p=p, seed=seed, drop_site=site, precision=prec(cfg)))
    pre = x3_ref(a, w.T) + bn + bm[:, None]
    keep = keep_mask(seed, site, M * N, p).reshape(M, N)
    np.testing.assert_allclose(be.host(CP), pre, atol=2e-5)
    np.testing.assert_allclose(be.host(C), 0.5 * pre * (1 + erf(pre / np.sqrt(2))) * keep / (1 - p) + r + c0, atol=4e-5)
    dy, w2, prea = f32(rng, M, K), f32(rng, K, N), f32(rng, M, N)
    DY, W2, PRE, C2 = be.dev(dy), be.dev(w2), be.dev(prea), be.zeros((M, N))
    run(be, mk(be, M, N, K, DY, D(K), D(1), W2, D(N), D(1), C2, D(N), D(1), R=be.ptr(PRE), Rm=D(N), Rn=D(1), act=_abi.ACT_GELU_GRAD,
               drop_p=p, seed=seed, drop_site=5, precision=prec(cfg)))
    keep = keep_mask(seed, 5, M * N, p).reshape(M, N)
    x = prea.astype(np.float64)
    gprime = 0.5 * (1 + erf(x / np.sqrt(2))) + x * np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)
    np.testing.assert_allclose(be.host(C2), x3_ref(dy, w2) * keep / (1 - p) * gprime, atol=4e-5)
    Bt, Cc, T = 3, 63, 50
    xe, we, b_, pe = f32(rng, Bt, Cc, T), f32(rng, T, T), f32(rng, T), f32(rng, Cc, T)
    X, WE, Bv, PE, OUT = be.dev(xe), be.dev(we), be.dev(b_), be.dev(pe), be.zeros((Bt, Cc + 1, T))
    d = mk(be, Bt * Cc, T, T, X, D(T), D(1), WE, D(1), D(T), OUT, D(T, div=Cc, so=(Cc + 1) * T), D(1), bias_n=be.ptr(Bv),
           R=be.ptr(PE), Rm=D(T, div=Cc, so=0), Rn=D(1), precision=prec(cfg))
    d.C = be.ptr(OUT) + T * 4
    run(be, d)
    out = be.host(OUT)
    np.testing.assert_allclose(out[:, 1:], x3_ref(xe.reshape(-1, T), we.T).reshape(Bt, Cc, T) + b_ + pe, atol=2e-5)
    assert (out[:, 0] == 0).all()


def test_x3_falls_back_to_f32_products_for_other_operand_classes(be):
    """odd sizes / two-level operand maps are outside the split kernel's operand class: the request is honoured with exact fp32 products"""
    rng = np.random.default_rng(9)
    M, N, K = 33, 35, 17
    a, b = f32(rng, M, K), f32(rng, K, N)
    A, B, C = be.dev(a), be.dev(b), be.zeros((M, N))
    run(be, mk(be, M, N, K, A, D(K), D(1), B, D(N), D(1), C, D(N), D(1), precision=prec()))
    np.testing.assert_allclose(be.host(C), a.astype(np.float64) @ b, atol=2e-5)
    d = mk(be, M, N, K, A, D(K), D(1), B, D(N), D(1), C, D(N), D(1), precision=7)
    assert be.lib.eegclip_gemm_f32(ctypes.byref(d), be.stream) < 0


@pytest.mark.parametrize("cfg", [0, 2, 3, 5])
@pytest.mark.parametrize("split_k", [1, 3])
def test_x3_value_embedding_weight_gradient_view(be, cfg, split_k):
    """dW = dOut[:, 1:, :]^T X with k = (sample, channel) through a two-level map on the gradient side (63 of every 64 token rows) and a
    plain one on the EEG side: the K2 form of the split kernel (Embed.py:146-149 backward), bias gradient as rowsum_a"""
    rng = np.random.default_rng(14 + split_k)
    Bt, Cc, Dm, T = 5, 63, 50, 70
    dout, x, w0 = f32(rng, Bt, Cc + 1, Dm), f32(rng, Bt, Cc, T), f32(rng, Dm, T)
    DO, X, W, RS = be.dev(dout), be.dev(x), be.dev(w0), be.zeros(Dm)
    d = mk(be, Dm, T, Bt * Cc, DO, D(1), D(Dm, div=Cc, so=(Cc + 1) * Dm), X, D(T), D(1), W, D(T), D(1), accumulate=1, split_k=split_k,
           rowsum_a=be.ptr(RS), precision=prec(cfg))
    d.A = be.ptr(DO) + 4 * Dm                     # skip token row 0 of sample 0
    run(be, d)
    g = dout[:, 1:, :].reshape(-1, Dm)
    np.testing.assert_allclose(be.host(W), w0 + x3_ref(g.T, x.reshape(-1, T)), atol=2e-4)        # fp32 accumulation over K = 315
    np.testing.assert_allclose(be.host(RS), g.astype(np.float64).sum(0), atol=2e-4)


def planes_of(be, w, transpose=False):
    """eegclip_split_rows on one matrix -> (hi, lo, ld) device planes + the numpy reference (hi + lo as float64, padded)"""
    rows, cols = w.shape
    orow, ocol = (cols, rows) if transpose else (rows, cols)
    ld = (ocol + 63) // 64 * 64
    W = be.dev(w)
    hi, lo = be.dev(np.full((orow, ld), 0x7FC0, np.uint16)), be.dev(np.full((orow, ld), 0x7FC0, np.uint16))
    it = (_abi.SplitItem * 1)(_abi.SplitItem(src=be.ptr(W), hi=be.ptr(hi), lo=be.ptr(lo), rows=rows, cols=cols, ld_src=cols, ld_out=ld, transpose=int(transpose)))
    assert be.lib.eegclip_split_rows(it, 1, be.stream) == 0
    return hi, lo, ld, W


def test_split_rows_planes_and_padding(be):
    rng = np.random.default_rng(1)
    w = f32(rng, 37, 50)
    for tr in (False, True):
        hi, lo, ld, _ = planes_of(be, w, tr)
        src = w.T if tr else w
        h = (be.host(hi).astype(np.uint32) << 16).view(np.float32)
        l = (be.host(lo).astype(np.uint32) << 16).view(np.float32)
        assert ld == 64 and h.shape == (src.shape[0], 64)
        np.testing.assert_array_equal(h[:, :src.shape[1]], bf16_round(src))
        np.testing.assert_array_equal(l[:, :src.shape[1]], bf16_round((src - bf16_round(src)).astype(np.float32)))
        assert (h[:, src.shape[1]:] == 0).all() and (l[:, src.shape[1]:] == 0).all()


def test_split_rows_also_stacks_fp32_copies(be):
    """eegclip_split_item.copy: the launch that splits several matrices into planes also leaves their fp32 rows in ONE stacked matrix (the image and text
    targets as one operand of the loss's query gradient); both kernel paths (dense 16-byte, element-wise with a padded ld), refused with transpose"""
    rng = np.random.default_rng(3)
    for rows, cols, ld_copy in ((32, 64, 64), (5, 50, 72)):
        ws = [f32(rng, rows, cols) for _ in range(3)]
        ld = (cols + 63) // 64 * 64 if cols % 64 else cols
        W = [be.dev(w) for w in ws]
        hi = [be.dev(np.zeros((rows, ld), np.uint16)) for _ in ws]
        lo = [be.dev(np.zeros((rows, ld), np.uint16)) for _ in ws]
        stack = be.dev(np.full((2 * rows, ld_copy), 7.0, np.float32))
        its = (_abi.SplitItem * 3)()
        for i in range(3):
            its[i] = _abi.SplitItem(src=be.ptr(W[i]), hi=be.ptr(hi[i]), lo=be.ptr(lo[i]), rows=rows, cols=cols, ld_src=cols, ld_out=ld, transpose=0,
                                    copy=be.ptr(stack) + 4 * (i - 1) * rows * ld_copy if i else None, ld_copy=ld_copy)
        assert be.lib.eegclip_split_rows(its, 3, be.stream) == 0
        got = be.host(stack)
        np.testing.assert_array_equal(got[:rows, :cols], ws[1])
        np.testing.assert_array_equal(got[rows:, :cols], ws[2])
        assert (got[:, cols:] == 7.0).all()                                      # the padding of the copy is not touched
        for i in range(3):
            h = (be.host(hi[i]).astype(np.uint32) << 16).view(np.float32)
            np.testing.assert_array_equal(h[:, :cols], bf16_round(ws[i]))
    its[1].transpose = 1
    assert be.lib.eegclip_split_rows(its, 3, be.stream) != 0
    its[1].transpose, its[1].ld_copy = 0, 8
    assert be.lib.eegclip_split_rows(its, 3, be.stream) != 0


@pytest.mark.parametrize("cfg", [0, 2, 3, 5])
@pytest.mark.parametrize("M,N,K,split_k", [(72, 66, 330, 3), (200, 130, 70, 2), (130, 250, 1100, 8), (2, 2, 40, 5)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1)])
def test_x3_split_k_through_the_workspace(be, cfg, M, N, K, split_k, ta, tb):
    """split-K with eegclip_gemm_desc.workspace: every slice writes its partial product into its own slab, a second kernel adds the slabs in
    slice order.  Same sums as the atomic route, bit-equal from run to run, bias / alpha / accumulate / bias_m applied once, workspace contents
    irrelevant on entry (NaN filled here)."""
    if be.name == "emu" and cfg in (2, 3) and (M, N, K) != (72, 66, 330):
        pytest.skip("emulator time")
    rng = np.random.default_rng(M + N + K + 5 * ta + 3 * tb)
    a = f32(rng, *((K, M) if ta else (M, K)))
    b = f32(rng, *((N, K) if tb else (K, N)))
    bias, bm, c0 = f32(rng, N), f32(rng, M), f32(rng, M, N)
    A, B, BI, BM = be.dev(a), be.dev(b), be.dev(bias), be.dev(bm)
    Am, Ak = (D(1), D(M)) if ta else (D(K), D(1))
    Bk, Bn = (D(1), D(K)) if tb else (D(N), D(1))
    outs = []
    ws = None
    for rep in range(2):
        C = be.dev(c0)
        d = mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), bias_n=be.ptr(BI), bias_m=be.ptr(BM), alpha=0.5, accumulate=1, split_k=split_k,
               precision=prec(cfg))
        need = be.lib.eegclip_gemm_workspace_bytes(ctypes.byref(d))
        assert need == 4 * split_k * M * ((N + 3) // 4 * 4)
        if ws is None:
            ws = be.dev(np.full(need // 4, np.nan, np.float32))
        d.workspace, d.workspace_bytes = be.ptr(ws), need
        run(be, d)
        outs.append(be.host(C).copy())
    am, bmat = (a.T if ta else a), (b.T if tb else b)
    want = 0.5 * x3_ref(am, bmat) + bias + bm[:, None] + c0
    np.testing.assert_allclose(outs[0], want, atol=6e-6 * max(1.0, np.abs(want).max()))
    assert (outs[0] == outs[1]).all()
    # too small a workspace: the atomic route, same sums
    C = be.dev(c0)
    d = mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), bias_n=be.ptr(BI), bias_m=be.ptr(BM), alpha=0.5, accumulate=1, split_k=split_k,
           precision=prec(cfg))
    d.workspace, d.workspace_bytes = be.ptr(ws), need - 4
    run(be, d)
    np.testing.assert_allclose(be.host(C), want, atol=6e-6 * max(1.0, np.abs(want).max()))


def test_x3_split_k_workspace_query_is_zero_where_no_workspace_is_used(be):
    rng = np.random.default_rng(3)
    a, b = f32(rng, 40, 30), f32(rng, 30, 20)
    A, B, C = be.dev(a), be.dev(b), be.zeros((40, 20))
    for split_k, precision, want_ws in [(1, prec(), False), (4, 0, False), (4, prec(), True)]:
        d = mk(be, 40, 20, 30, A, D(30), D(1), B, D(20), D(1), C, D(20), D(1), accumulate=1, split_k=split_k, precision=precision)
        assert (be.lib.eegclip_gemm_workspace_bytes(ctypes.byref(d)) > 0) == want_ws


@pytest.mark.parametrize("split_k", [3, 7])
def test_x3_value_embedding_weight_gradient_through_the_workspace(be, split_k):
    rng = np.random.default_rng(40 + split_k)
    Bt, Cc, Dm, T = 5, 63, 50, 70
    dout, x, w0 = f32(rng, Bt, Cc + 1, Dm), f32(rng, Bt, Cc, T), f32(rng, Dm, T)
    DO, X, W, RS = be.dev(dout), be.dev(x), be.dev(w0), be.zeros(Dm)
    d = mk(be, Dm, T, Bt * Cc, DO, D(1), D(Dm, div=Cc, so=(Cc + 1) * Dm), X, D(T), D(1), W, D(T), D(1), accumulate=1, split_k=split_k,
           rowsum_a=be.ptr(RS), precision=prec())
    d.A = be.ptr(DO) + 4 * Dm
    need = be.lib.eegclip_gemm_workspace_bytes(ctypes.byref(d))
    assert need > 0
    ws = be.zeros(need // 4)
    d.workspace, d.workspace_bytes = be.ptr(ws), need
    run(be, d)
    g = dout[:, 1:, :].reshape(-1, Dm)
    np.testing.assert_allclose(be.host(W), w0 + x3_ref(g.T, x.reshape(-1, T)), atol=2e-4)
    np.testing.assert_allclose(be.host(RS), g.astype(np.float64).sum(0), atol=2e-4)


@pytest.mark.parametrize("prec_", [0, _abi.PREC_BF16X3])
@pytest.mark.parametrize("N", [92, 250])
def test_gemm_accumulate_first_then_dropout(be, prec_, N):
    """accumulate = 2: C = dropout(alpha A B + bias + C_old) -- the dropout backward of a sum of two gradient contributions as the epilogue of the
    GEMM that produces the second one (rows of 250 floats start off the 4-element Philox blocks: two blocks per lane)"""
    rng = np.random.default_rng(7 + N)
    M, K = 70, 40
    a, w, bn, c0 = f32(rng, M, K), f32(rng, K, N), f32(rng, N), f32(rng, M, N)
    A, W, BN, C = be.dev(a), be.dev(w), be.dev(bn), be.dev(c0)
    p, seed, site = 0.25, 0x0BADC0FFEE123457, 6
    run(be, mk(be, M, N, K, A, D(K), D(1), W, D(N), D(1), C, D(N), D(1), bias_n=be.ptr(BN), accumulate=2, drop_p=p, seed=seed, drop_site=site, precision=prec_))
    keep = keep_mask(seed, site, M * N, p).reshape(M, N)
    prod = x3_ref(a, w) if prec_ else a.astype(np.float64) @ w.astype(np.float64)
    np.testing.assert_allclose(be.host(C), (prod + bn + c0) * keep / (1 - p), atol=4e-5)
    d = mk(be, M, N, K, A, D(K), D(1), W, D(N), D(1), C, D(N), D(1), accumulate=2, split_k=2, precision=prec_)
    assert be.lib.eegclip_gemm_f32(ctypes.byref(d), be.stream) < 0          # (no accumulate-first across K slices)
