"""csrc/head_gemm.hip through the C ABI on both backends: the K-parallel plane GEMM of the projection head (slabs + tickets, last arriver runs the epilogue)
against numpy on the SAME split operands (three products, fp64 accumulation).  Covers every epilogue the plans use (bias + pre-activation copy + GELU +
planes; GELU' with the residual in place; plain) at slices = 1, slab outputs at 2 .. 16 slices incl. uneven k ranges, M / N edges (rows and columns past
the end are never stored), and the launches that add the slabs while they load them: eegclip_head_act(_bwd), the slab forms of the head's LayerNorm launches
against their plain forms bit for bit, the transposing weight split at a column count that is not a multiple of 64."""
import numpy as np
import pytest
from scipy.special import erf

from backends import be  # noqa: F401
from eeg_image_decode_amd import _abi
from test_kernels_gemm_planes import from_planes, planes, reference
from test_kernels_wgrad import split


def gelu(x):
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def gelu_grad(x):
    return 0.5 * (1.0 + erf(x / np.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / np.sqrt(2.0 * np.pi)


def run(be, a, b, slices, opts, guard=8):
    M, K = a.shape
    N = b.shape[0]
    rng = np.random.default_rng(7 * M + N + K)
    (ah, al), _ = planes(be, a)
    km = int(bool(opts.get("kmajor")))                                      # B handed over as B[k][n] planes (what dX = dY W reads: the forward's weight planes)
    (bh, bl), _ = planes(be, np.ascontiguousarray(b.T)) if km else planes(be, b)
    ldb = N if km else K
    ldc = N + (guard if opts.get("strided") else 0)
    want = reference(a, b)
    if slices > 1:
        # K-parallel: slab s of C holds slice s's partial tile; nothing else may be requested and nothing past an edge is stored
        stride = (M + 1) * ldc + 4
        C = be.dev(np.full(slices * stride, np.nan, np.float32))
        d = _abi.HeadGemmDesc(a_hi=be.ptr(ah), a_lo=be.ptr(al), b_hi=be.ptr(bh), b_lo=be.ptr(bl), lda=K, ldb=ldb, M=M, N=N, K=K, slices=slices,
                              slab_stride=stride, C=be.ptr(C), ldc=ldc, b_kmajor=km)
        for _ in range(opts.get("calls", 1)):
            assert be.lib.eegclip_head_gemm(d, be.stream) == 0
        be.sync()
        got = be.host(C).reshape(slices, stride)
        slabs = got[:, :(M + 1) * ldc].reshape(slices, M + 1, ldc)
        assert np.isnan(got[:, (M + 1) * ldc:]).all() and np.isnan(slabs[:, M]).all() and np.isnan(slabs[:, :M, N:]).all()
        total = slabs[0, :M, :N].copy()
        for sl in range(1, slices):
            total = total + slabs[sl, :M, :N]                               # fp32, slice order: what the consumer kernels compute
        np.testing.assert_allclose(total, want, atol=4e-6 * max(1.0, float(np.abs(want).max())), rtol=2e-6)
        # every slab is the split-product contraction over ITS k range
        kt = K // 32
        for sl in range(slices):
            k0, k1 = 32 * (kt * sl // slices), 32 * (kt * (sl + 1) // slices)
            np.testing.assert_allclose(slabs[sl, :M, :N], reference(a[:, k0:k1], b[:, k0:k1]), atol=4e-6 * max(1.0, float(np.abs(want).max())), rtol=2e-6)
        return {"C": total, "slabs": slabs[:, :M, :N].copy()}
    bias = rng.standard_normal(N).astype(np.float32)
    aux = rng.standard_normal((M, ldc)).astype(np.float32)
    r0 = rng.standard_normal((M + 1, ldc)).astype(np.float32)              # (one guard row behind the matrix: must stay untouched)
    C = be.dev(r0.copy()) if opts.get("inplace") else be.dev(np.full((M + 1, ldc), np.nan, np.float32))
    R = C if opts.get("inplace") else be.dev(r0.copy())
    Cpre = be.dev(np.full((M + 1, ldc), np.nan, np.float32))
    ph, plo = be.dev(np.full((M + 1, ldc), 0x7FC0, np.uint16)), be.dev(np.full((M + 1, ldc), 0x7FC0, np.uint16))
    BIAS, AUX = be.dev(bias), be.dev(aux)
    act = opts.get("act", 0)
    d = _abi.HeadGemmDesc(a_hi=be.ptr(ah), a_lo=be.ptr(al), b_hi=be.ptr(bh), b_lo=be.ptr(bl), lda=K, ldb=ldb, M=M, N=N, K=K, slices=1, slab_stride=0, b_kmajor=km,
                          bias=be.ptr(BIAS) if opts.get("bias") else None, Cpre=be.ptr(Cpre) if opts.get("cpre") else None, ldcpre=ldc, act=act,
                          aux=be.ptr(AUX) if act == _abi.ACT_GELU_GRAD else None, ldaux=ldc, R=be.ptr(R) if opts.get("R") or opts.get("inplace") else None,
                          ldr=ldc, C=None if opts.get("only_planes") else be.ptr(C), ldc=ldc, p_hi=be.ptr(ph) if opts.get("planes") else None,
                          p_lo=be.ptr(plo) if opts.get("planes") else None, ldp=ldc)
    assert be.lib.eegclip_head_gemm(d, be.stream) == 0
    be.sync()
    if opts.get("bias"):
        want = want + bias
    pre = want.copy()
    if act == _abi.ACT_GELU:
        want = gelu(want)
    elif act == _abi.ACT_GELU_GRAD:
        want = want * gelu_grad(aux[:M, :N].astype(np.float64))
    if opts.get("R") or opts.get("inplace"):
        want = want + r0[:M, :N]
    tol = 4e-6 * max(1.0, float(np.abs(pre).max()))
    out = {}
    if not opts.get("only_planes"):
        got = be.host(C)
        np.testing.assert_allclose(got[:M, :N], want, atol=tol, rtol=2e-6)
        if not opts.get("inplace"):
            assert np.isnan(got[M]).all() and np.isnan(got[:M, N:]).all()  # nothing stored past an edge
        else:
            np.testing.assert_array_equal(got[M], r0[M])
            np.testing.assert_array_equal(got[:M, N:], r0[:M, N:])
        out["C"] = got[:M, :N].copy()
    if opts.get("cpre"):
        np.testing.assert_allclose(be.host(Cpre)[:M, :N], pre, atol=tol, rtol=2e-6)
        assert np.isnan(be.host(Cpre)[M]).all()
    if opts.get("planes"):
        hi, lo = from_planes(be.host(ph)[:M, :N], be.host(plo)[:M, :N])
        np.testing.assert_allclose(hi.astype(np.float64) + lo, want, atol=tol + 2e-5 * np.abs(want).max(), rtol=2e-5)
        if "C" in out:                                                     # the planes ARE the split of the stored fp32 value
            h2, l2 = split(out["C"])
            np.testing.assert_array_equal(hi, h2)
            np.testing.assert_array_equal(lo, l2)
        assert (be.host(ph)[M] == 0x7FC0).all()
    return out


@pytest.mark.parametrize("M,N,K,slices,opts", [
    (64, 64, 32, 1, dict()),
    (64, 64, 96, 1, dict(bias=1)),
    (128, 128, 160, 1, dict(bias=1, cpre=1, act=_abi.ACT_GELU, planes=1)),                # slices = 1: the launch runs the epilogue itself
    (64, 192, 224, 1, dict(act=_abi.ACT_GELU_GRAD, inplace=1, planes=1)),                 # du = ds + dgu * gelu'(u), in place
    (100, 96, 64, 1, dict(bias=1, strided=1)),                                            # rows past M: clamped loads, no stores
    (64, 104, 128, 1, dict(strided=1, planes=1)),                                         # N = 104: a 40-column edge tile
    (64, 64, 512, 1, dict(only_planes=1, planes=1)),
    (64, 64, 96, 3, dict()),                                                              # K-parallel: one slab per slice
    (128, 128, 160, 2, dict(calls=2)),
    (100, 96, 64, 2, dict(strided=1)),
    (64, 104, 128, 4, dict(strided=1)),                                                   # (the 1440-column input gradient's edge tile)
    (8, 72, 64, 2, dict()),                                                               # a batch smaller than a tile
    (64, 64, 512, 16, dict()),
    (192, 64, 352, 7, dict()),                                                            # 11 k-tiles over 7 slices: uneven ranges
    (64, 64, 64, 1, dict(kmajor=1)),                                                      # B as B[k][n] planes: fragments through the LDS transpose read
    (128, 192, 160, 2, dict(kmajor=1)),
    (100, 104, 128, 4, dict(kmajor=1, strided=1)),                                        # ... with an edge tile in M and in N (N % 8 == 0)
    (64, 128, 96, 1, dict(kmajor=1, bias=1, cpre=1, act=_abi.ACT_GELU, planes=1)),
])
def test_head_gemm_against_split_products(be, M, N, K, slices, opts):
    rng = np.random.default_rng(M + N + K)
    a, b = rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    run(be, a, b, slices, opts)


def test_head_gemm_is_bit_reproducible_and_slice_count_changes_only_rounding(be):
    rng = np.random.default_rng(3)
    M, N, K = 128, 128, 256
    a, b = rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    first = run(be, a, b, 4, dict())["slabs"]
    for _ in range(2):
        np.testing.assert_array_equal(run(be, a, b, 4, dict())["slabs"], first)
    np.testing.assert_allclose(run(be, a, b, 8, dict())["C"], run(be, a, b, 1, dict())["C"], atol=8e-6)


def test_head_gemm_argument_checks(be):
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((64, 64)).astype(np.float32), rng.standard_normal((64, 64)).astype(np.float32)
    (ah, al), _ = planes(be, a)
    (bh, bl), _ = planes(be, b)
    C = be.zeros((2, 64, 64))
    base = dict(a_hi=be.ptr(ah), a_lo=be.ptr(al), b_hi=be.ptr(bh), b_lo=be.ptr(bl), lda=64, ldb=64, M=64, N=64, K=64, slices=1, slab_stride=0,
                bias=None, Cpre=None, ldcpre=64, act=0, aux=None, ldaux=64, R=None, ldr=64, C=be.ptr(C), ldc=64, p_hi=None, p_lo=None, ldp=64)
    call = lambda **kw: be.lib.eegclip_head_gemm(_abi.HeadGemmDesc(**dict(base, **kw)), be.stream)
    assert call() == 0
    assert call(slices=2, slab_stride=4096) == 0
    assert call(slices=2, slab_stride=4000) != 0                           # slabs would overlap
    assert call(slices=2, slab_stride=4096, bias=be.ptr(C)) != 0           # no epilogue with slabs: the consumer runs it
    assert call(slices=2, slab_stride=4096, act=_abi.ACT_GELU) != 0
    assert call(slices=3, slab_stride=4096) != 0                           # more slices than k-tiles
    assert call(K=48) != 0 and call(N=62) != 0
    assert call(b_kmajor=1) == 0 and call(b_kmajor=1, N=60, ldb=60) != 0   # k-major B: N % 8 == 0
    assert call(act=_abi.ACT_GELU_GRAD) != 0                               # needs aux
    assert call(act=_abi.ACT_SILU) != 0
    assert call(C=None) != 0                                               # no output at all
    assert call(p_hi=be.ptr(C)) != 0                                       # one plane without the other
    be.sync()
    sl = be.lib.eegclip_head_gemm_slices
    assert (int(sl(256, 1024, 1440)), int(sl(256, 1024, 1024)), int(sl(256, 1440, 1024)), int(sl(256, 1024, 512)), int(sl(8, 64, 64))) == (4, 4, 2, 4, 1)


# ------------------------------------------------------------------------------------------------ the consumers that add the slabs
def _slabs(rng, n, M, N, pad=8):
    stride = M * N + pad
    buf = np.full(n * stride, np.nan, np.float32)
    parts = rng.standard_normal((n, M, N)).astype(np.float32)
    for s_ in range(n):
        buf[s_ * stride:s_ * stride + M * N] = parts[s_].ravel()
    total = parts[0].copy()
    for s_ in range(1, n):
        total = total + parts[s_]
    return buf, stride, total


@pytest.mark.parametrize("n", [1, 4])
def test_head_act_and_its_backward_add_the_slabs(be, n):
    rng = np.random.default_rng(10 + n)
    M, N = 24, 72
    buf, stride, x = _slabs(rng, n, M, N)
    bias = rng.standard_normal(N).astype(np.float32)
    S, BIAS = be.dev(buf), be.dev(bias)
    pre, out = be.dev(np.full((M, N), np.nan, np.float32)), be.dev(np.full((M, N), np.nan, np.float32))
    ph, plo = be.zeros((M, N), np.uint16), be.zeros((M, N), np.uint16)
    assert be.lib.eegclip_head_act(be.ptr(S), n, stride, be.ptr(BIAS), be.ptr(pre), be.ptr(out), be.ptr(ph), be.ptr(plo), M, N, be.stream) == 0
    be.sync()
    u = x + bias
    np.testing.assert_array_equal(be.host(pre), u)                         # fp32 adds in slice order, then the bias: bit for bit
    np.testing.assert_allclose(be.host(out), gelu(u.astype(np.float64)), atol=2e-6, rtol=2e-6)
    hi, lo = from_planes(be.host(ph), be.host(plo))
    h2, l2 = split(be.host(out))
    np.testing.assert_array_equal(hi, h2)
    np.testing.assert_array_equal(lo, l2)
    # backward: dx = base + slabs * gelu'(pre), in place on base
    base0 = rng.standard_normal((M, N)).astype(np.float32)
    DX = be.dev(base0.copy())
    assert be.lib.eegclip_head_act_bwd(be.ptr(S), n, stride, be.ptr(pre), be.ptr(DX), be.ptr(DX), be.ptr(ph), be.ptr(plo), M * N, be.stream) == 0
    be.sync()
    want = base0 + x.astype(np.float64) * gelu_grad(u.astype(np.float64))
    np.testing.assert_allclose(be.host(DX), want, atol=3e-6, rtol=3e-6)
    hi, lo = from_planes(be.host(ph), be.host(plo))
    h2, l2 = split(be.host(DX))
    np.testing.assert_array_equal(hi, h2)
    np.testing.assert_array_equal(lo, l2)
    assert be.lib.eegclip_head_act(be.ptr(S), 2, 8, be.ptr(BIAS), be.ptr(pre), None, None, None, M, N, be.stream) != 0      # slabs would overlap
    assert be.lib.eegclip_head_act(be.ptr(S), 1, 0, None, None, None, None, None, M, N, be.stream) != 0                    # nothing to write


@pytest.mark.parametrize("n,p", [(1, 0.0), (3, 0.0), (4, 0.5)])
def test_layernorm_launches_add_the_slabs(be, n, p):
    """eegclip_residual_layernorm_fwd_slabs == eegclip_residual_layernorm_fwd_planes on the summed operand; eegclip_layernorm_bwd_slabs == eegclip_layernorm_bwd
    on the summed gradient, with dropout'(dx) also as planes"""
    rng = np.random.default_rng(20 + n)
    R_, Cc = 12, 1024
    buf, stride, x = _slabs(rng, n, R_, Cc)
    bias = rng.standard_normal(Cc).astype(np.float32)
    resid = rng.standard_normal((R_, Cc)).astype(np.float32)
    g, b_ = rng.standard_normal(Cc).astype(np.float32), rng.standard_normal(Cc).astype(np.float32)
    S, BIAS, RES, G, Bt = be.dev(buf), be.dev(bias), be.dev(resid), be.dev(g), be.dev(b_)
    xsum = be.dev(x + bias)
    outs = []
    for slabs in (True, False):
        xo, y = be.dev(np.full((R_, Cc), np.nan, np.float32)), be.dev(np.full((R_, Cc), np.nan, np.float32))
        mu, rs = be.zeros(R_), be.zeros(R_)
        yh, yl = be.zeros((R_, Cc), np.uint16), be.zeros((R_, Cc), np.uint16)
        if slabs:
            rc = be.lib.eegclip_residual_layernorm_fwd_slabs(be.ptr(S), be.ptr(RES), be.ptr(xo), p, 77, 6, be.ptr(G), be.ptr(Bt), be.ptr(y), be.ptr(mu), be.ptr(rs),
                                                             None, None, None, None, None, R_, Cc, 1e-5, be.ptr(yh), be.ptr(yl), n, stride, be.ptr(BIAS), be.stream)
        else:
            rc = be.lib.eegclip_residual_layernorm_fwd_planes(be.ptr(xsum), be.ptr(RES), be.ptr(xo), p, 77, 6, be.ptr(G), be.ptr(Bt), be.ptr(y), be.ptr(mu), be.ptr(rs),
                                                              None, None, None, None, None, R_, Cc, 1e-5, be.ptr(yh), be.ptr(yl), be.stream)
        assert rc == 0
        be.sync()
        outs.append([be.host(t_) for t_ in (xo, y, mu, rs, yh, yl)])
    for a_, b2 in zip(*outs):
        np.testing.assert_array_equal(a_, b2)
    # backward
    xo, mu, rs = outs[0][0], outs[0][2], outs[0][3]
    dbuf, dstride, dy = _slabs(rng, n, R_, Cc)
    DS, DY, XO, MU, RS = be.dev(dbuf), be.dev(dy), be.dev(xo), be.dev(mu), be.dev(rs)
    res = []
    for slabs in (True, False):
        dx, dd = be.dev(np.full((R_, Cc), np.nan, np.float32)), be.dev(np.full((R_, Cc), np.nan, np.float32))
        dh, dl = be.zeros((R_, Cc), np.uint16), be.zeros((R_, Cc), np.uint16)
        dysum = be.dev(np.full((R_, Cc), np.nan, np.float32))
        if slabs:
            rc = be.lib.eegclip_layernorm_bwd_slabs(be.ptr(DS), n, dstride, be.ptr(XO), be.ptr(G), be.ptr(MU), be.ptr(RS), be.ptr(dx), R_, Cc, be.ptr(dd), be.ptr(dh),
                                                    be.ptr(dl), be.ptr(dysum), p, 77, 6, be.stream)
        else:
            rc = be.lib.eegclip_layernorm_bwd(be.ptr(DY), be.ptr(XO), be.ptr(G), be.ptr(MU), be.ptr(RS), be.ptr(dx), None, None, R_, Cc, 0, be.ptr(dd), p, 77, 6, be.stream)
        assert rc == 0
        be.sync()
        res.append((be.host(dx), be.host(dd), be.host(dh), be.host(dl)))
        if slabs:
            np.testing.assert_array_equal(be.host(dysum), dy)              # the summed gradient for the parameter half
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    hi, lo = from_planes(res[0][2], res[0][3])
    h2, l2 = split(res[0][1])
    np.testing.assert_array_equal(hi, h2)
    np.testing.assert_array_equal(lo, l2)


def test_split_transpose_takes_a_column_count_that_is_not_a_multiple_of_64(be):
    """the head's first weight (1024, 1440) -> planes of its transpose for the input-gradient GEMM: 1440 = 22.5 tiles of 64 columns"""
    rng = np.random.default_rng(5)
    rows, cols, ld_out = 128, 104, 136
    w = rng.standard_normal((rows, cols)).astype(np.float32)
    W = be.dev(w)
    hi, lo = be.dev(np.full((cols + 1, ld_out), 0x7FC0, np.uint16)), be.dev(np.full((cols + 1, ld_out), 0x7FC0, np.uint16))
    item = (_abi.SplitItem * 1)(_abi.SplitItem(src=be.ptr(W), hi=be.ptr(hi), lo=be.ptr(lo), rows=rows, cols=cols, ld_src=cols, ld_out=ld_out, transpose=1))
    assert be.lib.eegclip_split_transpose(item, 1, be.stream) == 0
    be.sync()
    h, l_ = from_planes(be.host(hi), be.host(lo))
    h2, l2 = split(np.ascontiguousarray(w.T))
    np.testing.assert_array_equal(h[:cols, :rows], h2)
    np.testing.assert_array_equal(l_[:cols, :rows], l2)
    assert (be.host(hi)[cols] == 0x7FC0).all() and (be.host(hi)[:cols, rows:] == 0x7FC0).all()      # nothing past the transposed matrix


def test_split_rows_lds_free_transposing_path_writes_column_blocks_without_padding(be):
    """eegclip_split_rows transpose = 2: two targets (n, D) -> the column blocks of ONE (D, 2 n) plane pair (the query-gradient GEMM's k-contiguous B operand),
    nothing written outside a block; a dense one-row item (the head's [W1 | b1 | W2] span) in the same launch"""
    rng = np.random.default_rng(6)
    n, Dm = 12, 40
    bs = [rng.standard_normal((n, Dm)).astype(np.float32) for _ in range(2)]
    span = rng.standard_normal(3 * 64).astype(np.float32)
    hi, lo = be.dev(np.full((Dm + 1, 2 * n), 0x7FC0, np.uint16)), be.dev(np.full((Dm + 1, 2 * n), 0x7FC0, np.uint16))
    sh, sl = be.zeros(span.size, np.uint16), be.zeros(span.size, np.uint16)
    B0, B1, SP = be.dev(bs[0]), be.dev(bs[1]), be.dev(span)
    items = (_abi.SplitItem * 3)(
        _abi.SplitItem(src=be.ptr(B0), hi=be.ptr(hi), lo=be.ptr(lo), rows=n, cols=Dm, ld_src=Dm, ld_out=2 * n, transpose=2),
        _abi.SplitItem(src=be.ptr(B1), hi=be.ptr(hi) + 2 * n, lo=be.ptr(lo) + 2 * n, rows=n, cols=Dm, ld_src=Dm, ld_out=2 * n, transpose=2),
        _abi.SplitItem(src=be.ptr(SP), hi=be.ptr(sh), lo=be.ptr(sl), rows=1, cols=span.size, ld_src=span.size, ld_out=span.size, transpose=0))
    assert be.lib.eegclip_split_rows(items, 3, be.stream) == 0
    be.sync()
    h, l_ = from_planes(be.host(hi), be.host(lo))
    want = np.concatenate([b.T for b in bs], axis=1)
    h2, l2 = split(np.ascontiguousarray(want))
    np.testing.assert_array_equal(h[:Dm], h2)
    np.testing.assert_array_equal(l_[:Dm], l2)
    assert (be.host(hi)[Dm] == 0x7FC0).all() and (be.host(lo)[Dm] == 0x7FC0).all()
    h, l_ = from_planes(be.host(sh), be.host(sl))
    h2, l2 = split(span)
    np.testing.assert_array_equal(h, h2)
    np.testing.assert_array_equal(l_, l2)
    bad = (_abi.SplitItem * 1)(_abi.SplitItem(src=be.ptr(B0), hi=be.ptr(hi), lo=be.ptr(lo), rows=n - 1, cols=Dm, ld_src=Dm, ld_out=2 * n, transpose=2))
    assert be.lib.eegclip_split_rows(bad, 1, be.stream) != 0               # rows % 4 != 0


@pytest.mark.parametrize("n,T,nslabs", [(64, 2, 3), (128, 1, 1), (24, 3, 2), (8, 1, 8)])
def test_infonce_small_against_fp64(be, n, T, nslabs):
    """csrc/infonce_small.hip: from partial slabs of the raw logits -> loss, d loss / d scale and the gradient matrices as planes, against an fp64 evaluation of
    models/loss.py:122-140 (symmetric cross-entropy of S = s A B^T, RAW scale) for T targets with weights w_t"""
    rng = np.random.default_rng(n + T)
    NC = T * n
    raw = (rng.standard_normal((n, NC)) * 4).astype(np.float32)
    parts = rng.standard_normal((nslabs, n, NC)).astype(np.float32)
    parts[nslabs - 1] = raw - parts[:nslabs - 1].sum(0)
    tot = parts[0].copy()
    for i in range(1, nslabs):
        tot = tot + parts[i]
    stride = n * NC + 16
    buf = np.full(nslabs * stride, np.nan, np.float32)
    for i in range(nslabs):
        buf[i * stride:i * stride + n * NC] = parts[i].ravel()
    s = np.float32(2.6593)
    w = [0.99, 0.01, 0.4, 0.7][:T] if T > 1 else [0.7]
    SL, SC = be.dev(buf), be.dev(np.array([s], np.float32))
    nws = int(be.lib.eegclip_infonce_small_workspace_floats(n, T))
    WS = be.dev(np.full(nws, np.nan, np.float32))
    GH, GL = be.dev(np.full((n, NC + 8), 0x7FC0, np.uint16)), be.dev(np.full((n, NC + 8), 0x7FC0, np.uint16))
    ACC = be.dev(np.array([0.5, -0.25], np.float32))
    assert be.lib.eegclip_infonce_small_fwd(be.ptr(SL), nslabs, stride, n, T, be.ptr(SC), be.ptr(WS), be.stream) == 0
    w4 = w + [0.0] * (4 - T)
    assert be.lib.eegclip_infonce_small_grad(n, T, be.ptr(SC), be.ptr(WS), *w4, be.ptr(GH), be.ptr(GL), NC + 8, be.ptr(ACC), be.ptr(ACC) + 4, be.stream) == 0
    be.sync()
    want_loss, want_ds = 0.0, 0.0
    G = np.zeros((n, NC))
    for t in range(T):
        R = tot[:, t * n:(t + 1) * n].astype(np.float64)
        S = float(s) * R
        lr = np.log(np.exp(S - S.max(1, keepdims=True)).sum(1)) + S.max(1)
        lc = np.log(np.exp(S - S.max(0, keepdims=True)).sum(0)) + S.max(0)
        d = np.diag(S)
        want_loss += w[t] * 0.5 / n * ((lr - d).sum() + (lc - d).sum())
        g = w[t] * 0.5 / n * (np.exp(S - lr[:, None]) + np.exp(S - lc[None, :]) - 2 * np.eye(n))
        want_ds += (g * R).sum()
        G[:, t * n:(t + 1) * n] = g * float(s)
    acc = be.host(ACC)
    assert abs(acc[0] - 0.5 - want_loss) < 2e-5 * max(1.0, abs(want_loss)), (acc[0] - 0.5, want_loss)
    assert abs(acc[1] + 0.25 - want_ds) < 1e-4 * max(1.0, abs(want_ds)), (acc[1] + 0.25, want_ds)
    hi, lo = from_planes(be.host(GH)[:, :NC], be.host(GL)[:, :NC])
    np.testing.assert_allclose(hi.astype(np.float64) + lo, G, atol=3e-5 * np.abs(G).max() + 1e-9)
    assert (be.host(GH)[:, NC:] == 0x7FC0).all()
    assert be.lib.eegclip_infonce_small_supported(100, 2) == 0 and be.lib.eegclip_infonce_small_supported(96, 2) == 1 and be.lib.eegclip_infonce_small_supported(64, 5) == 0 and be.lib.eegclip_infonce_small_supported(256, 2) == 1
