"""Kernel-logic tests of csrc/gemm.hip under the CPU lane emulator (tests/hipemu).  The same source is what hipcc
builds for gfx950; `-m gpu` tests repeat these checks on the real device through the product loader."""
import ctypes

import numpy as np
import pytest

from hipemu import emu
from eeg_image_decode_amd import _abi
from philox_np import keep_mask

pytestmark = pytest.mark.emu
D = _abi.dim


def run(desc):
    rc = emu.lib().eegclip_gemm_f32(ctypes.byref(desc), None)
    assert rc == 0, rc


def mk(M, N, K, A, Am, Ak, B, Bk, Bn, C, Cm, Cn, **kw):
    d = _abi.GemmDesc(M=M, N=N, K=K, A=emu.ptr(A), Am=Am, Ak=Ak, B=emu.ptr(B), Bk=Bk, Bn=Bn, C=emu.ptr(C), Cm=Cm, Cn=Cn,
                      Cpre=None, bias_n=None, bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=0, act=0,
                      drop_p=0.0, seed=0, drop_site=0, split_k=1)
    for k, v in kw.items():
        setattr(d, k, v)
    return d


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (64, 64, 32), (100, 70, 50), (63, 250, 250), (130, 33, 97)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts(M, N, K, ta, tb):
    rng = np.random.default_rng(M * 1000 + N * 10 + K + ta * 2 + tb)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    C = np.full((M, N), np.nan, np.float32)
    Am, Ak = (D(1), D(M)) if ta else (D(K), D(1))
    Bk, Bn = (D(1), D(K)) if tb else (D(N), D(1))
    run(mk(M, N, K, A, Am, Ak, B, Bk, Bn, C, D(N), D(1), alpha=0.5))
    a = A.T if ta else A
    b = B.T if tb else B
    ref = 0.5 * a.astype(np.float64) @ b.astype(np.float64)
    np.testing.assert_allclose(C, ref, atol=2e-5 * max(1, np.abs(ref).max()))


def test_gemm_epilogue_bias_gelu_pre_residual_accumulate():
    from scipy.special import erf
    rng = np.random.default_rng(5)
    M, N, K = 70, 90, 40
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    bn = rng.standard_normal(N).astype(np.float32)
    bm = rng.standard_normal(M).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    C = C0.copy()
    Cpre = np.zeros((M, N), np.float32)
    run(mk(M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1), Cpre=emu.ptr(Cpre), bias_n=emu.ptr(bn), bias_m=emu.ptr(bm),
           R=emu.ptr(R), Rm=D(N), Rn=D(1), act=_abi.ACT_GELU, accumulate=1))
    pre = A.astype(np.float64) @ W.T + bn + bm[:, None]
    ref = 0.5 * pre * (1 + erf(pre / np.sqrt(2))) + R + C0
    np.testing.assert_allclose(Cpre, pre, atol=2e-5)
    np.testing.assert_allclose(C, ref, atol=3e-5)


def test_gemm_dropout_mask_is_philox_of_logical_index():
    rng = np.random.default_rng(6)
    M, N, K = 65, 67, 8
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    C = np.zeros((M, N), np.float32)
    p, seed, site = 0.25, 0x1234567890ABCDEF, 3
    run(mk(M, N, K, A, D(K), D(1), W, D(1), D(K), C, D(N), D(1), drop_p=p, seed=seed, drop_site=site))
    keep = keep_mask(seed, site, M * N, p).reshape(M, N)
    ref = (A.astype(np.float64) @ W.T) * keep / (1 - p)
    np.testing.assert_allclose(C, ref, atol=2e-5)
    assert 0.6 < keep.mean() < 0.9


def test_gemm_split_k_accumulates_atomically():
    rng = np.random.default_rng(7)
    M, N, K = 40, 75, 1000
    A = rng.standard_normal((K, M)).astype(np.float32)      # dW = dY^T X : both operands reduce over rows
    B = rng.standard_normal((K, N)).astype(np.float32)
    bn = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    C = C0.copy()
    run(mk(M, N, K, A, D(1), D(M), B, D(N), D(1), C, D(N), D(1), split_k=5, bias_n=emu.ptr(bn), alpha=2.0))
    ref = C0 + 2.0 * A.T.astype(np.float64) @ B + bn
    np.testing.assert_allclose(C, ref, atol=3e-4)


def test_gemm_two_level_dims_embedding_view():
    """Value-embedding GEMM writes rows 1..63 of each (64,250) token block and adds PE[channel] (Embed.py:146-160)."""
    rng = np.random.default_rng(8)
    Bt, Cc, T = 3, 63, 50
    x = rng.standard_normal((Bt, Cc, T)).astype(np.float32)
    W = rng.standard_normal((T, T)).astype(np.float32)
    b = rng.standard_normal(T).astype(np.float32)
    pe = rng.standard_normal((Cc, T)).astype(np.float32)
    out = np.zeros((Bt, Cc + 1, T), np.float32)
    d = mk(Bt * Cc, T, T, x, D(T), D(1), W, D(1), D(T), out, D(T, div=Cc, so=(Cc + 1) * T), D(1), bias_n=emu.ptr(b),
           R=emu.ptr(pe), Rm=D(T, div=Cc, so=0), Rn=D(1))
    d.C = out.ctypes.data + T * 4            # skip token row 0 of batch 0
    run(d)
    ref = x.astype(np.float64) @ W.T + b + pe
    np.testing.assert_allclose(out[:, 1:], ref, atol=2e-5)
    assert (out[:, 0] == 0).all()


def test_gemm_spatial_conv_view():
    """(63x1) conv as one GEMM over the (B,40,63,36) tensor: M = out ch, N = (b,w) two-level, K = (c,h)."""
    rng = np.random.default_rng(9)
    Bt, Ci, H, Wd, Co = 3, 5, 7, 36, 6
    z = rng.standard_normal((Bt, Ci, H, Wd)).astype(np.float32)
    W2 = rng.standard_normal((Co, Ci, H)).astype(np.float32)
    bias = rng.standard_normal(Co).astype(np.float32)
    y = np.zeros((Bt, Co, Wd), np.float32)
    run(mk(Co, Bt * Wd, Ci * H, W2, D(Ci * H), D(1), z, D(Wd), D(1, div=Wd, so=Ci * H * Wd), y, D(Wd), D(1, div=Wd, so=Co * Wd),
           bias_m=emu.ptr(bias)))
    ref = np.einsum("och,bchw->bow", W2.astype(np.float64), z.astype(np.float64)) + bias[None, :, None]
    np.testing.assert_allclose(y, ref, atol=2e-5)


def test_gemm_rejects_bad_arguments():
    L = emu.lib()
    assert L.eegclip_gemm_f32(None, None) < 0
    A = np.zeros((4, 4), np.float32)
    d = mk(4, 4, 4, A, D(4), D(1), A, D(4), D(1), A.copy(), D(4), D(1), split_k=2, act=_abi.ACT_GELU)
    assert L.eegclip_gemm_f32(ctypes.byref(d), None) < 0       # act with split-K is not allowed
    d = mk(4, 4, 4, A, D(4), D(1), A, D(4), D(1), A.copy(), D(4), D(1), drop_p=1.0)
    assert L.eegclip_gemm_f32(ctypes.byref(d), None) < 0
