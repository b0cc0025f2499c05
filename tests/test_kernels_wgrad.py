"""csrc/wgrad_planes.hip through the C ABI on both backends: the transposing split (fp32 [t][c] -> bf16 planes [c][t]) and the planes GEMM
dW += dY^T X with its bias row sums, against float64 products of the split operands (the kernel's arithmetic) and of the exact operands."""
import numpy as np
import pytest

from backends import be  # noqa: F401
from test_kernels_gemm_x3 import split


def planes_of(be, x, out_rows, pad=0):
    rows, cols = x.shape
    X = be.dev(x)
    hi, lo = be.zeros((out_rows, rows + pad), np.uint16), be.zeros((out_rows, rows + pad), np.uint16)
    assert be.lib.eegclip_split_transpose(be.ptr(X), x.strides[0] // 4, rows, cols, out_rows, be.ptr(hi), be.ptr(lo), rows + pad, be.stream) == 0
    return X, hi, lo


def bf16_to_f64(u):
    return (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


@pytest.mark.parametrize("rows,cols", [(64, 62), (128, 250), (192, 130)])
def test_split_transpose(be, rows, cols):
    rng = np.random.default_rng(rows + cols)
    x = rng.standard_normal((rows, cols + 3)).astype(np.float32)[:, :cols]          # a strided view: ld != cols
    xs = np.ascontiguousarray(x)
    out_rows = (cols + 63) // 64 * 64
    _, hi, lo = planes_of(be, xs, out_rows)
    h, l = bf16_to_f64(be.host(hi))[:, :rows], bf16_to_f64(be.host(lo))[:, :rows]
    xh, xl = split(xs)
    np.testing.assert_array_equal(h[:cols], xh.T)
    np.testing.assert_array_equal(l[:cols], xl.T)
    assert not h[cols:].any() and not l[cols:].any()


@pytest.mark.parametrize("M,N,K,bias", [(250, 256, 256, True), (62, 40, 128, False), (300, 250, 512, True)])
def test_wgrad_planes_matches_the_split_products(be, M, N, K, bias):
    rng = np.random.default_rng(M + N + K)
    dy = (rng.standard_normal((K, M)) * rng.uniform(0.1, 2.0, M)).astype(np.float32)
    x = rng.standard_normal((K, N)).astype(np.float32)
    Mp, Np = (M + 127) // 128 * 128, (N + 63) // 64 * 64
    _, ah, al = planes_of(be, dy, Mp, 64)                                              # plane rows K + 64 apart, as the plans allocate them
    _, bh, bl = planes_of(be, x, Np, 64)
    c0 = rng.standard_normal((M, N + 2)).astype(np.float32)                           # accumulated INTO, row stride N + 2
    b0 = rng.standard_normal(M).astype(np.float32)
    C, Bv = be.dev(c0), be.dev(b0)
    ws = be.zeros(int(be.lib.eegclip_wgrad_planes_workspace_floats(M, N, K)))
    assert be.lib.eegclip_wgrad_planes(be.ptr(ah), be.ptr(al), be.ptr(bh), be.ptr(bl), K + 64, M, N, K, be.ptr(C), N + 2, be.ptr(Bv) if bias else None, be.ptr(ws),
                                       be.stream) == 0
    dh, dl = split(dy)
    xh, xl = split(x)
    want = dh.T @ xh + dh.T @ xl + dl.T @ xh                                          # the three products of the split arithmetic
    got = be.host(C)
    np.testing.assert_allclose(got[:, :N] - c0[:, :N], want, atol=2e-5 * np.abs(want).max() + 1e-6)
    np.testing.assert_array_equal(got[:, N:], c0[:, N:])
    exact = dy.astype(np.float64).T @ x.astype(np.float64)
    assert np.abs(want - exact).max() < 1e-4 * max(1.0, np.abs(exact).max())        # split products vs exact ones: the parity budget
    if bias:
        np.testing.assert_allclose(be.host(Bv) - b0, (dh + dl).sum(0), atol=2e-5 * np.abs(dy).sum(0).max())


def natural_planes(be, x, ldp):
    rows, cols = x.shape
    X = be.dev(np.ascontiguousarray(x))
    hi, lo = be.dev(np.full((rows, ldp), 0x7FC0, np.uint16)), be.dev(np.full((rows, ldp), 0x7FC0, np.uint16))
    assert be.lib.eegclip_split_rows_natural(be.ptr(X), cols, rows, cols, be.ptr(hi), be.ptr(lo), ldp, be.stream) == 0
    return X, hi, lo


@pytest.mark.parametrize("rows,cols", [(64, 62), (96, 250), (32, 744)])
def test_split_rows_natural(be, rows, cols):
    rng = np.random.default_rng(rows * 3 + cols)
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    ldp = (cols + 7) // 8 * 8
    _, hi, lo = natural_planes(be, x, ldp)
    xh, xl = split(x)
    np.testing.assert_array_equal(bf16_to_f64(be.host(hi))[:, :cols], xh)
    np.testing.assert_array_equal(bf16_to_f64(be.host(lo))[:, :cols], xl)
    assert not bf16_to_f64(be.host(hi))[:, cols:].any() and not bf16_to_f64(be.host(lo))[:, cols:].any()


@pytest.mark.parametrize("M,N,K,bias", [(250, 256, 256, True), (62, 40, 64, False), (744, 250, 128, True), (300, 130, 512, True)])
def test_wgrad_tr_from_natural_planes_matches_the_split_products(be, M, N, K, bias):
    """dW += dY^T X with both operands as token-major planes, fetched through the LDS transpose read (a transpose-detecting check: M != N, random data)"""
    rng = np.random.default_rng(M + 2 * N + K)
    dy = (rng.standard_normal((K, M)) * rng.uniform(0.1, 2.0, M)).astype(np.float32)
    x = rng.standard_normal((K, N)).astype(np.float32)
    lda, ldb = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    _, ah, al = natural_planes(be, dy, lda)
    _, bh, bl = natural_planes(be, x, ldb)
    c0 = rng.standard_normal((M, N + 2)).astype(np.float32)                           # accumulated INTO, row stride N + 2
    b0 = rng.standard_normal(M).astype(np.float32)
    C, Bv = be.dev(c0), be.dev(b0)
    ws = be.dev(np.full(int(be.lib.eegclip_wgrad_tr_workspace_floats(M, N, K)), np.nan, np.float32))
    assert be.lib.eegclip_wgrad_tr(be.ptr(ah), be.ptr(al), lda, be.ptr(bh), be.ptr(bl), ldb, M, N, K, be.ptr(C), N + 2, be.ptr(Bv) if bias else None,
                                   be.ptr(ws), be.stream) == 0
    dh, dl = split(dy)
    xh, xl = split(x)
    want = dh.T @ xh + dh.T @ xl + dl.T @ xh
    got = be.host(C)
    np.testing.assert_allclose(got[:, :N] - c0[:, :N], want, atol=2e-5 * np.abs(want).max() + 1e-6)
    np.testing.assert_array_equal(got[:, N:], c0[:, N:])
    if bias:
        np.testing.assert_allclose(be.host(Bv) - b0, (dh + dl).sum(0), atol=2e-5 * np.abs(dy).sum(0).max())
    assert be.lib.eegclip_wgrad_tr(be.ptr(ah), be.ptr(al), lda, be.ptr(bh), be.ptr(bl), ldb, M, N, K + 8, be.ptr(C), N + 2, None, be.ptr(ws), be.stream) < 0
