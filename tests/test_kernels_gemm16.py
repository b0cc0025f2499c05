"""csrc/gemm16.hip through the C ABI on both backends: the 16-bit Linear of the SDXL sampling path (fp16 / bf16, bias, residual shaped like the
output or one row per sample, ragged M) and the fused sampler step."""
import numpy as np
import pytest

from backends import be  # noqa: F401
from eeg_image_decode_amd import _abi
from test_kernels_gemm_x3 import bf16_round

DT = {"f16": _abi.DT_F16, "bf16": _abi.DT_BF16}


def to16(x, dt):
    """fp32 array -> (uint16 bit pattern, the value it represents as fp32)"""
    if dt == "f16":
        h = x.astype(np.float16)
        return h.view(np.uint16), h.astype(np.float32)
    v = bf16_round(x)
    return (v.view(np.uint32) >> 16).astype(np.uint16), v


def from16(u, dt):
    u = np.asarray(u, np.uint16)
    if dt == "f16":
        return u.view(np.float16).astype(np.float32)
    return (u.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("M,N,K,r_div,use_bias", [(128, 128, 64, 0, True), (200, 256, 192, 0, True), (77, 128, 128, -1, False), (96, 128, 64, 32, True),
                                                  (300, 128, 256, 100, False)])
def test_gemm16_linear_bias_residual(be, dt, M, N, K, r_div, use_bias):
    rng = np.random.default_rng(M + N + K)
    a16, a = to16(rng.standard_normal((M, K)).astype(np.float32), dt)
    w16, w = to16((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32), dt)
    b16, b = to16(rng.standard_normal(N).astype(np.float32), dt)
    rrows = M if r_div == 0 else (M + r_div - 1) // r_div if r_div > 0 else 0
    A, W, B = be.dev(a16), be.dev(w16), be.dev(b16)
    C = be.dev(np.full((M, N), 0x7E00, np.uint16))
    R = r = None
    if rrows:
        r16, r = to16(rng.standard_normal((rrows, N)).astype(np.float32), dt)
        R = be.dev(r16)
    rc = be.lib.eegclip_gemm16(be.ptr(A), K, be.ptr(W), K, be.ptr(C), N, be.ptr(B) if use_bias else None, be.ptr(R), N, max(r_div, 0), M, N, K, DT[dt],
                               be.stream)
    assert rc == 0
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    if use_bias:
        ref = ref + b
    if rrows:
        ref = ref + (r if r_div == 0 else r[np.arange(M) // r_div])
    got = from16(be.host(C), dt)
    ulp = 2.0 ** -10 if dt == "f16" else 2.0 ** -7               # output rounding: half an ulp of the 16-bit result, plus fp32 accumulation
    np.testing.assert_allclose(got, ref, atol=ulp * np.abs(ref).max() * 0.6 + 1e-3)


def test_gemm16_rejects_unsupported_shapes(be):
    z = be.zeros((128, 128), np.uint16)
    L = be.lib
    assert L.eegclip_gemm16(be.ptr(z), 128, be.ptr(z), 128, be.ptr(z), 128, None, None, 0, 0, 128, 100, 128, 0, be.stream) < 0      # N % 128
    assert L.eegclip_gemm16(be.ptr(z), 128, be.ptr(z), 128, be.ptr(z), 128, None, None, 0, 0, 128, 128, 96, 0, be.stream) < 0       # K % 64
    assert L.eegclip_gemm16(be.ptr(z), 128, be.ptr(z), 128, be.ptr(z), 128, None, None, 0, 0, 128, 128, 128, 5, be.stream) < 0      # dtype
    assert L.eegclip_gemm16(be.ptr(z), 128, be.ptr(z), 128, be.ptr(z), 128, None, None, 0, 0, 0, 128, 128, 0, be.stream) == 0       # M = 0: nothing to do


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("cfg,with_noise", [(True, False), (False, True), (True, True)])
def test_sampler_step_cfg_mix_and_linear_update(be, dt, cfg, with_noise):
    rng = np.random.default_rng(3)
    n = 4 * 1000
    x16, x = to16(rng.standard_normal(n).astype(np.float32), dt)
    u16, u = to16(rng.standard_normal(n).astype(np.float32), dt)
    c16, c = to16(rng.standard_normal(n).astype(np.float32), dt)
    z16, z = to16(rng.standard_normal(n).astype(np.float32), dt)
    X, U, Cc, Z = be.dev(x16), be.dev(u16), be.dev(c16), be.dev(z16)
    OUT, SC = be.zeros(n, np.uint16), be.zeros(n, np.uint16)
    g, cx, ce, cn, ins = 5.0, 0.98, -0.11, 0.07, 0.83
    rc = be.lib.eegclip_sampler_step(be.ptr(X), be.ptr(U), be.ptr(Cc) if cfg else None, be.ptr(Z) if with_noise else None, be.ptr(OUT), be.ptr(SC), g, cx, ce,
                                     cn, ins, n, DT[dt], be.stream)
    assert rc == 0
    eps = u + g * (c - u) if cfg else u
    ref = cx * x + ce * eps + (cn * z if with_noise else 0)
    _, ref16 = to16(ref.astype(np.float32), dt)
    got = from16(be.host(OUT), dt)
    ulp = 2.0 ** -10 if dt == "f16" else 2.0 ** -7
    np.testing.assert_allclose(got, ref16, atol=ulp * np.abs(ref).max())          # (fp32 vs fp64 evaluation may round to the neighbouring 16-bit value)
    _, sref = to16((got * ins).astype(np.float32), dt)
    np.testing.assert_allclose(from16(be.host(SC), dt), sref, atol=ulp * np.abs(ref).max())
