"""csrc/gemm.hip through the C ABI on both backends (CPU lane emulator; MI355X with -m gpu): all four operand layouts,
ragged sizes, two-level index maps (the encoder's strided views), every epilogue stage, split-K, argument errors."""
import ctypes

import numpy as np
import pytest

from backends import be  # noqa: F401
from eeg_image_decode_amd import _abi
from philox_np import keep_mask

D = _abi.dim


def run(be, desc):
    rc = be.lib.eegclip_gemm_f32(ctypes.byref(desc), be.stream)
    assert rc == 0, rc


def mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, Cm, Cn, **kw):
    d = _abi.GemmDesc(M=M, N=N, K=K, A=be.ptr(A), Am=Am, Ak=Ak, B=be.ptr(B), Bk=Bk, Bn=Bn, C=be.ptr(C), Cm=Cm, Cn=Cn,
                      Cpre=None, bias_n=None, bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=0, act=0,
                      drop_p=0.0, seed=0, drop_site=0, split_k=1)
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def f32(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (64, 64, 32), (100, 70, 50), (63, 250, 250), (130, 33, 97)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts(be, M, N, K, ta, tb):
    rng = np.random.default_rng(M * 1000 + N * 10 + K + ta * 2 + tb)
    a = f32(rng, *((K, M) if ta else (M, K)))
    b = f32(rng, *((N, K) if tb else (K, N)))
    A, B, C = be.dev(a), be.dev(b), be.dev(np.full((M, N), np.nan, np.float32))
    Am, Ak = (D(1), D(M)) if ta else (D(K), D(1))
    Bk, Bn = (D(1), D(K)) if tb else (D(N), D(1))
    run(be, mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), alpha=0.5))
    ref = 0.5 * (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    np.testing.assert_allclose(be.host(C), ref, atol=2e-5 * max(1, np.abs(ref).max()))


def test_gemm_epilogue_bias_gelu_pre_residual_accumulate(be):
    from scipy.special import erf
    rng = np.random.default_rng(5)
    M, N, K = 70, 90, 40
    a, w, bn, bm, r, c0 = f32(rng, M, K), f32(rng, N, K), f32(rng, N), f32(rng, M), f32(rng, M, N), f32(rng, M, N)
    A, W, BN, BM, R, C, CP = be.dev(a), be.dev(w), be.dev(bn), be.dev(bm), be.dev(r), be.dev(c0), be.zeros((M, N))
    run(be, mk(be, M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1), Cpre=be.ptr(CP), bias_n=be.ptr(BN), bias_m=be.ptr(BM),
               R=be.ptr(R), Rm=D(N), Rn=D(1), act=_abi.ACT_GELU, accumulate=1))
    pre = a.astype(np.float64) @ w.T + bn + bm[:, None]
    ref = 0.5 * pre * (1 + erf(pre / np.sqrt(2))) + r + c0
    np.testing.assert_allclose(be.host(CP), pre, atol=2e-5)
    np.testing.assert_allclose(be.host(C), ref, atol=3e-5)


@pytest.mark.parametrize("M,N,K", [(65, 67, 8), (70, 72, 8), (130, 256, 34), (33, 4, 6)])
def test_gemm_dropout_mask_is_philox_of_logical_index(be, M, N, K):
    """(N % 4 == 0 takes the quad-shared Philox route of the epilogue, other N the per-element one: the mask must be the same function of
    the logical element index either way)"""
    rng = np.random.default_rng(6 + N)
    a, w = f32(rng, M, K), f32(rng, N, K)
    A, W, C = be.dev(a), be.dev(w), be.zeros((M, N))
    p, seed, site = 0.25, 0x1234567890ABCDEF, 3
    run(be, mk(be, M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1), drop_p=p, seed=seed, drop_site=site))
    keep = keep_mask(seed, site, M * N, p).reshape(M, N)
    np.testing.assert_allclose(be.host(C), (a.astype(np.float64) @ w.T) * keep / (1 - p), atol=2e-5)
    assert 0.6 < keep.mean() < 0.9


def test_gemm_split_k_accumulates_atomically(be):
    rng = np.random.default_rng(7)
    M, N, K = 40, 75, 1000
    a, b, bn, c0 = f32(rng, K, M), f32(rng, K, N), f32(rng, N), f32(rng, M, N)      # dW = dY^T X
    A, B, BN, C = be.dev(a), be.dev(b), be.dev(bn), be.dev(c0)
    run(be, mk(be, M, N, K, A, D(1), D(M), B, D(N), D(1), C, D(N), D(1), split_k=5, bias_n=be.ptr(BN), alpha=2.0))
    ref = c0 + 2.0 * a.T.astype(np.float64) @ b + bn
    np.testing.assert_allclose(be.host(C), ref, atol=3e-4)


def test_gemm_two_level_dims_embedding_view(be):
    """Value-embedding GEMM writes rows 1..63 of each (64,T) token block and adds PE[channel] (Embed.py:146-160)."""
    rng = np.random.default_rng(8)
    Bt, Cc, T = 3, 63, 50
    x, w, b, pe = f32(rng, Bt, Cc, T), f32(rng, T, T), f32(rng, T), f32(rng, Cc, T)
    X, W, Bv, PE, OUT = be.dev(x), be.dev(w), be.dev(b), be.dev(pe), be.zeros((Bt, Cc + 1, T))
    d = mk(be, Bt * Cc, T, T, X, D(T), D(1), W, D(1), D(T), OUT, D(T, div=Cc, so=(Cc + 1) * T), D(1), bias_n=be.ptr(Bv),
           R=be.ptr(PE), Rm=D(T, div=Cc, so=0), Rn=D(1))
    d.C = be.ptr(OUT) + T * 4            # skip token row 0 of sample 0
    run(be, d)
    out = be.host(OUT)
    np.testing.assert_allclose(out[:, 1:], x.astype(np.float64) @ w.T + b + pe, atol=2e-5)
    assert (out[:, 0] == 0).all()


def test_gemm_spatial_conv_view(be):
    """(63x1) conv as one GEMM over the (B,40,63,36) tensor: M = out ch, N = (b,w) two-level, K = (c,h)."""
    rng = np.random.default_rng(9)
    Bt, Ci, H, Wd, Co = 3, 5, 7, 36, 6
    z, w2, bias = f32(rng, Bt, Ci, H, Wd), f32(rng, Co, Ci, H), f32(rng, Co)
    Z, W2, BI, Y = be.dev(z), be.dev(w2), be.dev(bias), be.zeros((Bt, Co, Wd))
    run(be, mk(be, Co, Bt * Wd, Ci * H, W2, D(Ci * H), D(1), Z, D(Wd), D(1, div=Wd, so=Ci * H * Wd), Y, D(Wd),
               D(1, div=Wd, so=Co * Wd), bias_m=be.ptr(BI)))
    ref = np.einsum("och,bchw->bow", w2.astype(np.float64), z.astype(np.float64)) + bias[None, :, None]
    np.testing.assert_allclose(be.host(Y), ref, atol=2e-5)


def test_gemm_big_tile_grid(be):
    """A shape with many tiles in both directions and a long K (proj head 1440 -> 1024 at B = 96)."""
    rng = np.random.default_rng(10)
    M, N, K = 96, 1024, 1440
    a, w = f32(rng, M, K), f32(rng, N, K) * 0.03
    A, W, C = be.dev(a), be.dev(w), be.zeros((M, N))
    run(be, mk(be, M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1)))
    ref = a.astype(np.float64) @ w.T.astype(np.float64)
    np.testing.assert_allclose(be.host(C), ref, atol=1e-4)


@pytest.mark.parametrize("M,N,K,ta,tb", [(1800, 1790, 9, 0, 1), (1800, 1790, 9, 1, 0), (6200, 250, 7, 0, 1), (6200, 250, 7, 0, 0), (6200, 250, 7, 1, 1)])
def test_gemm_large_tile_variants(be, M, N, K, ta, tb):
    """shapes whose grids select the 128x128 / 128x64 workgroup tiles (>= 192 workgroups), ragged edges included"""
    rng = np.random.default_rng(M + N + K + ta * 2 + tb)
    a = f32(rng, *((K, M) if ta else (M, K)))
    b = f32(rng, *((N, K) if tb else (K, N)))
    bias = f32(rng, N)
    A, B, BI, C = be.dev(a), be.dev(b), be.dev(bias), be.dev(np.full((M, N), np.nan, np.float32))
    Am, Ak = (D(1), D(M)) if ta else (D(K), D(1))
    Bk, Bn = (D(1), D(K)) if tb else (D(N), D(1))
    run(be, mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), bias_n=be.ptr(BI)))
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    np.testing.assert_allclose(be.host(C), ref, atol=3e-5)


@pytest.mark.parametrize("M,N,K,split", [(2, 2, 2, 1), (130, 70, 50, 1), (64, 128, 96, 1), (250, 744, 250, 1), (72, 66, 330, 3)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_fast_path(be, M, N, K, split, ta, tb):
    """even sizes + plain strides select gemm_f32_fast_kernel (8-byte staging, clamped edges, XCD tile order) in all four operand
    layouts; the k tail (K % 32 != 0), ragged tile edges and split-K slices must behave exactly like the general kernel"""
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    a = f32(rng, *((K, M) if ta else (M, K)))
    b = f32(rng, *((N, K) if tb else (K, N)))
    bias, c0 = f32(rng, N), f32(rng, M, N)
    A, B, BI, C = be.dev(a), be.dev(b), be.dev(bias), be.dev(c0 if split > 1 else np.full((M, N), np.nan, np.float32))
    Am, Ak = (D(1), D(M)) if ta else (D(K), D(1))
    Bk, Bn = (D(1), D(K)) if tb else (D(N), D(1))
    run(be, mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), bias_n=be.ptr(BI), split_k=split))
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias + (c0 if split > 1 else 0)
    np.testing.assert_allclose(be.host(C), ref, atol=3e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("M,N,K,split", [(130, 70, 50, 1), (72, 66, 330, 3), (63, 33, 97, 2)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_rowsum_a_is_the_bias_gradient(be, M, N, K, split, ta, tb):
    """rowsum_a[m] += sum_k A[m,k], added once per K slice by the n = 0 tiles (fast and general kernels, all layouts)"""
    rng = np.random.default_rng(M + N + K + ta * 2 + tb)
    a = f32(rng, *((K, M) if ta else (M, K)))
    b = f32(rng, *((N, K) if tb else (K, N)))
    r0 = f32(rng, M)
    A, B, C, RS = be.dev(a), be.dev(b), be.zeros((M, N)), be.dev(r0)
    Am, Ak = (D(1), D(M)) if ta else (D(K), D(1))
    Bk, Bn = (D(1), D(K)) if tb else (D(N), D(1))
    run(be, mk(be, M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), split_k=split, accumulate=int(split > 1), rowsum_a=be.ptr(RS)))
    am = (a.T if ta else a).astype(np.float64)
    np.testing.assert_allclose(be.host(C), am @ (b.T if tb else b).astype(np.float64), atol=1e-4)
    np.testing.assert_allclose(be.host(RS), r0 + am.sum(1), atol=1e-4)


def test_gemm_gelu_grad_epilogue(be):
    """dX = dropout-mask(dY W) * gelu'(pre): the FFN activation backward fused into the GEMM (Transformer_EncDec.py:48)"""
    from scipy.special import erf
    rng = np.random.default_rng(12)
    M, N, K = 70, 90, 40
    dy, w, pre = f32(rng, M, K), f32(rng, K, N), f32(rng, M, N)
    DY, W, PRE, C = be.dev(dy), be.dev(w), be.dev(pre), be.zeros((M, N))
    p, seed, site = 0.25, 0x1234567890ABCDEF, 5
    run(be, mk(be, M, N, K, DY, D(K), D(1), W, D(N), D(1), C, D(N), D(1), R=be.ptr(PRE), Rm=D(N), Rn=D(1), act=_abi.ACT_GELU_GRAD,
               drop_p=p, seed=seed, drop_site=site))
    keep = keep_mask(seed, site, M * N, p).reshape(M, N)
    x = pre.astype(np.float64)
    gprime = 0.5 * (1 + erf(x / np.sqrt(2))) + x * np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)
    np.testing.assert_allclose(be.host(C), (dy.astype(np.float64) @ w) * keep / (1 - p) * gprime, atol=3e-5)
    d = mk(be, M, N, K, DY, D(K), D(1), W, D(N), D(1), C, D(N), D(1), act=_abi.ACT_GELU_GRAD)
    assert be.lib.eegclip_gemm_f32(ctypes.byref(d), be.stream) < 0      # needs the pre-activation in R


@pytest.mark.parametrize("split", [1, 3])
def test_gemm_value_embedding_weight_gradient_view(be, split):
    """dW = dOut[:, 1:, :]^T X: the contraction index k = (sample, channel) runs through a two-level map on the gradient side (63 of
    every 64 token rows) and a plain one on the EEG side -- the K2 instantiation of the fast kernel (Embed.py:146-149 backward)"""
    rng = np.random.default_rng(14 + split)
    Bt, Cc, Dm, T = 5, 63, 50, 70
    dout, x, w0 = f32(rng, Bt, Cc + 1, Dm), f32(rng, Bt, Cc, T), f32(rng, Dm, T)
    DO, X, W, RS = be.dev(dout), be.dev(x), be.dev(w0), be.zeros(Dm)
    d = mk(be, Dm, T, Bt * Cc, DO, D(1), D(Dm, div=Cc, so=(Cc + 1) * Dm), X, D(T), D(1), W, D(T), D(1), accumulate=1, split_k=split,
           rowsum_a=be.ptr(RS))
    d.A = be.ptr(DO) + 4 * Dm                     # skip token row 0 of sample 0
    run(be, d)
    g = dout[:, 1:, :].reshape(-1, Dm).astype(np.float64)
    np.testing.assert_allclose(be.host(W), w0 + g.T @ x.reshape(-1, T), atol=2e-4)
    np.testing.assert_allclose(be.host(RS), g.sum(0), atol=2e-4)


def _grouped(be, descs, n=None):
    arr = (_abi.GemmDesc * len(descs))(*descs)
    rc = be.lib.eegclip_gemm_f32_grouped(arr, len(descs) if n is None else n, be.stream)
    assert rc == 0, rc


@pytest.mark.parametrize("counts", [[2, 1, 4], [1, 1], [3], [1] * 16, [1] * 17])
def test_grouped_gemm_per_subject_value_embedding(be, counts):
    """SURVEY 8f row 1 (models/subject_layers/Embed.py:142-144): a subject-ordered batch, one Linear per subject.  The grouped launch must
    equal the member-by-member launches bit for bit (split_k = 1: no atomics) and the fp64 reference; 17 members exceed the group table and
    take the member-by-member route inside the library.  Forward view: rows = (sample, channel) -> token rows 1..63 of (B,64,D) + bias + PE."""
    rng = np.random.default_rng(100 + len(counts))
    Cc, Dm, T = 63, 50, 70
    Bt, S = sum(counts), len(counts)
    x, w, bias, pe = f32(rng, Bt, Cc, T), f32(rng, S, Dm, T), f32(rng, S, Dm), f32(rng, Cc, Dm)
    X, W, BI, PE = be.dev(x), be.dev(w), be.dev(bias), be.dev(pe)
    outs = []
    for grouped in (False, True):
        H = be.dev(np.full((Bt, Cc + 1, Dm), 7.0, np.float32))
        ds, st = [], 0
        for s, n in enumerate(counts):
            d = mk(be, n * Cc, Dm, T, X, D(T), D(1), W, D(1), D(T), H, D(Dm, div=Cc, so=(Cc + 1) * Dm), D(1), R=be.ptr(PE),
                   Rm=D(Dm, div=Cc, so=0), Rn=D(1))
            d.A, d.B, d.bias_n = be.ptr(X) + 4 * st * Cc * T, be.ptr(W) + 4 * s * Dm * T, be.ptr(BI) + 4 * s * Dm
            d.C = be.ptr(H) + 4 * (st * (Cc + 1) * Dm + Dm)
            ds.append(d)
            st += n
        if grouped:
            _grouped(be, ds)
        else:
            for d in ds:
                run(be, d)
        outs.append(be.host(H))
    assert np.array_equal(outs[0], outs[1])
    st = 0
    for s, n in enumerate(counts):
        ref = x[st:st + n].astype(np.float64) @ w[s].T + bias[s] + pe
        np.testing.assert_allclose(outs[1][st:st + n, 1:], ref, atol=1e-4)
        st += n
    assert (outs[1][:, 0] == 7.0).all()                               # token row 0 is not the GEMM's to write


@pytest.mark.parametrize("splits", [[1, 1, 1], [2, 1, 3]])
def test_grouped_gemm_per_subject_weight_gradients(be, splits):
    """backward of the above: dW_s += dOut_s[:, 1:, :]^T X_s with the bias gradient as rowsum_a -- K2 instantiation, per-member K and split_k"""
    rng = np.random.default_rng(7 + sum(splits))
    Cc, Dm, T = 63, 50, 70
    counts = [2, 5, 1]
    Bt, S = sum(counts), len(counts)
    dout, x, w0 = f32(rng, Bt, Cc + 1, Dm), f32(rng, Bt, Cc, T), f32(rng, S, Dm, T)
    DO, X, W, RS = be.dev(dout), be.dev(x), be.dev(w0), be.zeros((S, Dm))
    ds, st = [], 0
    for s, n in enumerate(counts):
        d = mk(be, Dm, T, n * Cc, DO, D(1), D(Dm, div=Cc, so=(Cc + 1) * Dm), X, D(T), D(1), W, D(T), D(1), accumulate=1, split_k=splits[s],
               rowsum_a=be.ptr(RS) + 4 * s * Dm)
        d.A, d.B, d.C = be.ptr(DO) + 4 * (st * (Cc + 1) * Dm + Dm), be.ptr(X) + 4 * st * Cc * T, be.ptr(W) + 4 * s * Dm * T
        ds.append(d)
        st += n
    _grouped(be, ds)
    st = 0
    for s, n in enumerate(counts):
        g = dout[st:st + n, 1:, :].reshape(-1, Dm).astype(np.float64)
        np.testing.assert_allclose(be.host(W)[s], w0[s] + g.T @ x[st:st + n].reshape(-1, T), atol=2e-4)
        np.testing.assert_allclose(be.host(RS)[s], g.sum(0), atol=2e-4)
        st += n


def test_grouped_gemm_mixed_members_and_errors(be):
    """members that do not share a kernel instantiation (different N / layouts) still give the n-launch result; bad members are rejected
    before anything is launched"""
    rng = np.random.default_rng(3)
    a1, b1, a2, b2 = f32(rng, 40, 30), f32(rng, 30, 20), f32(rng, 16, 66), f32(rng, 10, 16)
    A1, B1, A2, B2, C1, C2 = be.dev(a1), be.dev(b1), be.dev(a2), be.dev(b2), be.zeros((40, 20)), be.zeros((66, 10))
    d1 = mk(be, 40, 20, 30, A1, D(30), D(1), B1, D(20), D(1), C1, D(20), D(1))
    d2 = mk(be, 66, 10, 16, A2, D(1), D(66), B2, D(1), D(16), C2, D(10), D(1))
    _grouped(be, [d1, d2])
    np.testing.assert_allclose(be.host(C1), a1.astype(np.float64) @ b1, atol=1e-4)
    np.testing.assert_allclose(be.host(C2), a2.T.astype(np.float64) @ b2.T, atol=1e-4)
    _grouped(be, [d1, d2], n=0)                                       # nothing to do
    bad = mk(be, 40, 20, 30, A1, D(30), D(1), B1, D(20), D(1), C1, D(20), D(1), split_k=0)
    arr = (_abi.GemmDesc * 2)(d1, bad)
    assert be.lib.eegclip_gemm_f32_grouped(arr, 2, be.stream) < 0
    assert be.lib.eegclip_gemm_f32_grouped(None, 2, be.stream) < 0


@pytest.mark.parametrize("M,N,K", [(1, 64, 64), (8, 1024, 1024), (16, 1000, 512), (17, 70, 68), (32, 512, 1000), (16, 5, 2048)])
def test_gemm_skinny_rows_kernel(be, M, N, K):
    """M <= 32 against k-contiguous weights (the prior's sampling chain, diffusion_prior.py:340-378): the 16-wave split-K-in-workgroup kernel,
    plain and with every epilogue stage (bias, SiLU, dropout, residual, accumulate, Cpre)"""
    rng = np.random.default_rng(M * 7 + N + K)
    a, w, bn, r, c0 = f32(rng, M, K), f32(rng, N, K), f32(rng, N), f32(rng, M, N), f32(rng, M, N)
    A, W, BN, R = be.dev(a), be.dev(w), be.dev(bn), be.dev(r)
    ref = a.astype(np.float64) @ w.T.astype(np.float64)
    C = be.dev(np.full((M, N), np.nan, np.float32))
    run(be, mk(be, M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1), alpha=0.5))
    np.testing.assert_allclose(be.host(C), 0.5 * ref, atol=3e-5 * max(1, np.abs(ref).max()))
    C, CP = be.dev(c0), be.zeros((M, N))
    p = 0.25
    run(be, mk(be, M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1), bias_n=be.ptr(BN), act=_abi.ACT_SILU, R=be.ptr(R), Rm=D(N), Rn=D(1),
               accumulate=1, Cpre=be.ptr(CP), drop_p=p, seed=77, drop_site=3))
    pre = ref + bn
    keep = keep_mask(77, 3, M * N, p).reshape(M, N)
    want = pre / (1 + np.exp(-pre)) * keep / (1 - p) + r + c0
    np.testing.assert_allclose(be.host(CP), pre, atol=3e-5 * max(1, np.abs(ref).max()))
    np.testing.assert_allclose(be.host(C), want, atol=5e-5 * max(1, np.abs(ref).max()))


def test_gemm_rejects_bad_arguments(be):
    L = be.lib
    assert L.eegclip_gemm_f32(None, be.stream) < 0
    A = be.zeros((4, 4))
    d = mk(be, 4, 4, 4, A, D(4), D(1), A, D(4), D(1), be.zeros((4, 4)), D(4), D(1), split_k=2, act=_abi.ACT_GELU)
    assert L.eegclip_gemm_f32(ctypes.byref(d), be.stream) < 0       # act with split-K is not allowed
    d = mk(be, 4, 4, 4, A, D(4), D(1), A, D(4), D(1), be.zeros((4, 4)), D(4), D(1), drop_p=1.0)
    assert L.eegclip_gemm_f32(ctypes.byref(d), be.stream) < 0
