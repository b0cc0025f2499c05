"""csrc/gemm_planes.hip through the C ABI on both backends: C = A B^T from bf16 hi | lo planes against numpy on the SAME split operands (three
products, fp64 accumulation), every epilogue stage the diffusion-prior plans use: bias, pre-activation copy, SiLU, residual, accumulate, plane output
of the result or of the pre-activation; strided operands (column blocks of a wide buffer)."""
import numpy as np
import pytest

from backends import be  # noqa: F401
from eeg_image_decode_amd import _abi
from test_kernels_wgrad import bf16_round, split            # float32 hi / lo (round to nearest even)


def planes(be, x, ld=None, col0=0):
    """fp32 (rows, K) -> device hi / lo planes with row stride ld (the operand sits at column col0 of a wider NaN-filled buffer)"""
    rows, K = x.shape
    ld = ld or K
    hi, lo = split(x)
    out = []
    for p in (hi, lo):
        buf = np.full((rows, ld), 0x7FC0, np.uint16)
        buf[:, col0:col0 + K] = (p.view(np.uint32) >> 16).astype(np.uint16)
        out.append(be.dev(buf))
    return out, 2 * col0


def from_planes(hi, lo):
    f = lambda u: (np.asarray(u).astype(np.uint32) << 16).view(np.float32)
    return f(hi), f(lo)


def reference(a, b):
    ah, al = (v.astype(np.float64) for v in split(a))
    bh, bl = (v.astype(np.float64) for v in split(b))
    return ah @ bh.T + ah @ bl.T + al @ bh.T


def silu(x):
    return x / (1.0 + np.exp(-x))


@pytest.mark.parametrize("M,N,K,opts", [
    (64, 64, 32, dict()),
    (64, 128, 64, dict(bias=1, act=1, cpre=1, planes_of=1)),                 # the prior's time-embedding hidden layer: SiLU(x W^T + b), pre-activation kept
    (128, 64, 96, dict(bias=1, R=1, accumulate=1, planes_of=1)),            # stage input: C += t_emb + x  (C holds the condition embedding)
    (128, 128, 160, dict(cpre=1, R=1, planes_of=2)),                        # dx GEMM: pure result kept (planes of it), + skip-branch gradient into C
    (64, 192, 64, dict(strided=1, bias=1)),                                 # operands / outputs are column blocks of wide buffers
    (192, 64, 224, dict(only_planes=1)),
])
def test_gemm_planes_against_split_products(be, M, N, K, opts):
    rng = np.random.default_rng(M + N + K)
    a, b = rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    strided = opts.get("strided", 0)
    (ah, al), aoff = planes(be, a, K + 64 if strided else None, 32 if strided else 0)
    (bh, bl), boff = planes(be, b, K + 8 if strided else None, 8 if strided else 0)
    ldc = N + 12 if strided else N
    c0 = rng.standard_normal((M, ldc)).astype(np.float32)
    C, Cpre = be.dev(c0.copy()), be.dev(np.full((M, ldc), np.nan, np.float32))
    bias = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, ldc)).astype(np.float32)
    BIAS, R = be.dev(bias), be.dev(r)
    ph, plo = be.dev(np.full((M, ldc), 0x7FC0, np.uint16)), be.dev(np.full((M, ldc), 0x7FC0, np.uint16))
    only_planes = opts.get("only_planes", 0)
    d = _abi.GemmPlanesDesc(a_hi=be.ptr(ah) + aoff, a_lo=be.ptr(al) + aoff, b_hi=be.ptr(bh) + boff, b_lo=be.ptr(bl) + boff,
                            lda=K + 64 if strided else K, ldb=K + 8 if strided else K, M=M, N=N, K=K,
                            C=None if only_planes else be.ptr(C), ldc=ldc, Cpre=be.ptr(Cpre) if opts.get("cpre") else None, ldcpre=ldc,
                            bias=be.ptr(BIAS) if opts.get("bias") else None, R=be.ptr(R) if opts.get("R") else None, ldr=ldc,
                            p_hi=be.ptr(ph), p_lo=be.ptr(plo), ldp=ldc, act=_abi.ACT_SILU if opts.get("act") else 0,
                            accumulate=opts.get("accumulate", 0), planes_of=1 if only_planes else opts.get("planes_of", 0))
    assert be.lib.eegclip_gemm_planes(d, be.stream) == 0
    be.sync()
    want = reference(a, b)
    if opts.get("bias"):
        want = want + bias
    pre = want.copy()
    if opts.get("act"):
        want = silu(want)
    if opts.get("R"):
        want = want + r[:, :N]
    if opts.get("accumulate"):
        want = want + c0[:, :N]
    tol = 3e-6 * max(1.0, float(np.abs(pre).max()))
    if not only_planes:
        got = be.host(C)
        np.testing.assert_allclose(got[:, :N], want, atol=tol)
        np.testing.assert_array_equal(got[:, N:], c0[:, N:])
    if opts.get("cpre"):
        np.testing.assert_allclose(be.host(Cpre)[:, :N], pre, atol=tol)
    po = 1 if only_planes else opts.get("planes_of", 0)
    if po:
        src = (want if po == 1 else pre).astype(np.float32)
        hi, lo = from_planes(be.host(ph)[:, :N], be.host(plo)[:, :N])
        # the planes are the split of the fp32 value the kernel stored: hi + lo reproduces it to 2^-16 relative
        np.testing.assert_allclose(hi.astype(np.float64) + lo, src, atol=tol + 2.0 ** -15 * np.abs(src).max())
        stored = None if only_planes else (be.host(C)[:, :N] if po == 1 else be.host(Cpre)[:, :N])
        if stored is not None:                               # exactly the split of the fp32 value next to it
            np.testing.assert_array_equal(hi, bf16_round(stored))
            np.testing.assert_array_equal(lo, bf16_round(stored - hi))
        assert (np.asarray(be.host(ph))[:, N:] == 0x7FC0).all()


def test_gemm_planes_rejects_bad_arguments(be):
    z = be.dev(np.zeros((64, 64), np.uint16))
    c = be.zeros((64, 64))
    ok = dict(a_hi=be.ptr(z), a_lo=be.ptr(z), b_hi=be.ptr(z), b_lo=be.ptr(z), lda=64, ldb=64, M=64, N=64, K=64, C=be.ptr(c), ldc=64)
    assert be.lib.eegclip_gemm_planes(_abi.GemmPlanesDesc(**ok), be.stream) == 0
    for bad in (dict(M=96), dict(K=48), dict(lda=60), dict(C=None), dict(planes_of=1), dict(accumulate=1, C=None, Cpre=be.ptr(c), ldcpre=64), dict(act=_abi.ACT_GELU)):
        assert be.lib.eegclip_gemm_planes(_abi.GemmPlanesDesc(**{**ok, **bad}), be.stream) != 0, bad


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1024, 1024, 1024), (1024, 2880, 512), (1024, 64, 128)])
def test_gemm_planes_prior_shapes(M, N, K):
    from backends import get
    test_gemm_planes_against_split_products(get("gpu"), M, N, K, dict(bias=1, act=1, cpre=1, planes_of=1))


# ---- the weight-gradient form over plain planes (csrc/wgrad_tok.hip: eegclip_wgrad_planes) ---------------------------------------------------------
def tn_case(be, rows, M, N, slices, bias, strided, seed):
    rng = np.random.default_rng(seed)
    dy, x = rng.standard_normal((rows, M)).astype(np.float32), rng.standard_normal((rows, N)).astype(np.float32)
    # the kernel reads whole 128-channel tiles: the operands sit in buffers padded to the next multiple of 128 channels (NaN there: must not reach out)
    pad = lambda c: (c + 127) // 128 * 128
    lda, ldb = (pad(M) + 128, pad(N) + 256) if strided else (pad(M), pad(N))
    (ah, al), aoff = planes(be, dy, lda, 128 if strided else 0)
    (bh, bl), boff = planes(be, x, ldb, 256 if strided else 0)
    ldo = N + 4
    o0, b0 = rng.standard_normal((M, ldo)).astype(np.float32), rng.standard_normal(M).astype(np.float32)
    out, bo = be.dev(o0.copy()), be.dev(b0.copy())
    p = (_abi.WgradPlanesProblem * 1)(_abi.WgradPlanesProblem(a_hi=be.ptr(ah) + aoff, a_lo=be.ptr(al) + aoff, lda=lda, b_hi=be.ptr(bh) + boff, b_lo=be.ptr(bl) + boff,
                                                             ldb=ldb, rows=rows, M=M, N=N, out=be.ptr(out), ldo=ldo, bias_out=be.ptr(bo) if bias else None, slices=slices))
    assert be.lib.eegclip_wgrad_planes(p, 1, be.stream) == 0
    be.sync()
    ah_, al_ = (v.astype(np.float64) for v in split(dy))
    bh_, bl_ = (v.astype(np.float64) for v in split(x))
    want = ah_.T @ bh_ + ah_.T @ bl_ + al_.T @ bh_
    got = be.host(out).astype(np.float64)
    np.testing.assert_allclose(got[:, :N] - o0[:, :N], want, atol=3e-6 * np.abs(want).max() * np.sqrt(rows) + 1e-5)
    np.testing.assert_array_equal(got[:, N:], o0[:, N:])
    gb = be.host(bo).astype(np.float64) - b0
    if bias:
        np.testing.assert_allclose(gb, (ah_ + al_).sum(0), atol=3e-6 * np.sqrt(rows) * np.abs(dy).sum(0).max() + 1e-5)
    else:
        np.testing.assert_array_equal(gb, 0.0)


@pytest.mark.parametrize("rows,M,N,slices,bias,strided", [(64, 64, 64, 1, 1, 0), (128, 128, 256, 1, 0, 0), (160, 192, 64, 2, 1, 1), (256, 64, 128, 4, 1, 0),
                                                          (96, 320, 128, 3, 1, 1)])
def test_wgrad_planes_against_split_products(be, rows, M, N, slices, bias, strided):
    tn_case(be, rows, M, N, slices, bias, strided, seed=rows + M + N)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,slices", [(1024, 1024, 4), (2880, 512, 2), (64, 128, 8), (512, 1024, 4)])
def test_wgrad_planes_prior_shapes(M, N, slices):
    from backends import get
    tn_case(get("gpu"), 1024, M, N, slices, 1, 0, seed=M + N)


def test_wgrad_planes_rejects_bad_arguments(be):
    z = be.dev(np.zeros((64, 128), np.uint16))
    o = be.zeros((64, 64))
    ok = dict(a_hi=be.ptr(z), a_lo=be.ptr(z), lda=128, b_hi=be.ptr(z), b_lo=be.ptr(z), ldb=128, rows=64, M=64, N=64, out=be.ptr(o), ldo=64, bias_out=None, slices=1)
    mk = lambda **kw: (_abi.WgradPlanesProblem * 1)(_abi.WgradPlanesProblem(**{**ok, **kw}))
    assert be.lib.eegclip_wgrad_planes(mk(), 1, be.stream) == 0
    for bad in (dict(rows=48), dict(slices=3), dict(slices=0), dict(lda=60), dict(ldo=62), dict(N=62), dict(out=None), dict(a_lo=None)):
        assert be.lib.eegclip_wgrad_planes(mk(**bad), 1, be.stream) != 0, bad
    many = (_abi.WgradPlanesProblem * 25)(*[_abi.WgradPlanesProblem(**ok)] * 25)
    assert be.lib.eegclip_wgrad_planes(many, 25, be.stream) != 0
