"""csrc/vae.hip through the C ABI on both backends (CPU lane emulator / MI355X): the implicit-GEMM convolution over padded NHWC frames (3 x 3, 1 x 1, fused
nearest-2x upsampling, the stride-2 downsampler, residual epilogue, partial tiles), the direct kernel of the 3 / 4 / 8-channel layers, GroupNorm (+ SiLU),
the row softmax and the latent sampling -- against numpy restatements of the torch ops AutoencoderKL is made of (Generation/custom_pipeline.py:421,
custom_pipeline_low_level.py:8-31)."""
import numpy as np
import pytest

from backends import be, ok  # noqa: F401
from eeg_image_decode_amd import _abi
from test_kernels_wgrad import bf16_round


def bf(x):
    return bf16_round(np.asarray(x, np.float32))


def u16(x):
    return (np.ascontiguousarray(x, np.float32).view(np.uint32) >> 16).astype(np.uint16)


def f32(u):
    return (np.asarray(u).astype(np.uint32) << 16).view(np.float32)


def frame(x, pad):
    """(N, C, H, W) fp32 (bf16-representable) -> padded NHWC uint16"""
    N, C, H, W = x.shape
    f = np.zeros((N, H + 2 * pad, W + 2 * pad, C), np.uint16)
    f[:, pad:pad + H, pad:pad + W, :] = u16(x.transpose(0, 2, 3, 1))
    return f


def conv_ref(x, w, b, stride=1, pads=(1, 1, 1, 1), up=False):
    """numpy conv2d, fp64; x (N, Cin, H, W), w (Cout, Cin, KS, KS); pads (top, left, bottom, right)"""
    x = x.astype(np.float64)
    if up:
        x = x.repeat(2, axis=2).repeat(2, axis=3)
    pt, pl, pb, pr = pads
    xp = np.pad(x, ((0, 0), (0, 0), (pt, pb), (pl, pr)))
    N, Cin, Hp, Wp = xp.shape
    Cout, _, KS, _ = w.shape
    Ho, Wo = (Hp - KS) // stride + 1, (Wp - KS) // stride + 1
    out = np.zeros((N, Cout, Ho, Wo))
    for ky in range(KS):
        for kx in range(KS):
            patch = xp[:, :, ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride]
            out += np.einsum("nchw,oc->nohw", patch, w[:, :, ky, kx].astype(np.float64))
    return out + b.astype(np.float64)[None, :, None, None]


@pytest.mark.parametrize("case", [
    dict(Cin=64, Cout=128, H=5, W=7),                                      # matrix-core path, one partial tile
    dict(Cin=64, Cout=128, H=4, W=4, KS=1, res=1),
    dict(Cin=64, Cout=128, H=3, W=4, up=1),
    dict(Cin=64, Cout=128, H=6, W=6, stride=2),
    dict(Cin=4, Cout=16, H=5, W=4),                                        # direct path
    dict(Cin=8, Cout=8, H=3, W=3, KS=1, out_pad=0),
])
def test_conv16(be, case):
    rng = np.random.default_rng(case["Cin"] + case["H"])
    Cin, Cout, H, W = case["Cin"], case["Cout"], case["H"], case["W"]
    KS, stride, up, out_pad = case.get("KS", 3), case.get("stride", 1), case.get("up", 0), case.get("out_pad", 1)
    N = 2
    x, w, b = bf(rng.standard_normal((N, Cin, H, W))), bf(rng.standard_normal((Cout, Cin, KS, KS)) / (Cin * KS * KS) ** 0.5), bf(rng.standard_normal(Cout))
    if up:
        ref, pads = conv_ref(x, w, b, up=True), (1, 1)
    elif stride == 2:
        ref, pads = conv_ref(x, w, b, stride=2, pads=(0, 0, 1, 1)), (0, 0)
    else:
        p = (KS - 1) // 2
        ref, pads = conv_ref(x, w, b, pads=(p, p, p, p)), (p, p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    res = bf(rng.standard_normal((N, Cout, Ho, Wo))) if case.get("res") else None
    if res is not None:
        ref = ref + res
    XIN, WP, B = be.dev(frame(x, 1)), be.dev(u16(w.transpose(0, 2, 3, 1).reshape(Cout, KS * KS, Cin))), be.dev(u16(b))
    OUT = be.dev(np.full((N, Ho + 2 * out_pad, Wo + 2 * out_pad, Cout), 0x7FC0, np.uint16))
    RES = be.dev(frame(res, out_pad)) if res is not None else None
    d = _abi.Conv16Desc(in_=be.ptr(XIN), W=be.ptr(WP), out=be.ptr(OUT), bias=be.ptr(B), residual=be.ptr(RES), N=N, Hi=H, Wi=W, Cin=Cin, in_pad=1, Ho=Ho, Wo=Wo,
                        Cout=Cout, out_pad=out_pad, KS=KS, stride=stride, pad_top=pads[0], pad_left=pads[1], upsample=up, dtype=_abi.DT_BF16)
    ok(be.lib.eegclip_conv16(d, be.stream))
    be.sync()
    got = be.host(OUT)
    inner = f32(got[:, out_pad:out_pad + Ho, out_pad:out_pad + Wo, :]).transpose(0, 3, 1, 2)
    np.testing.assert_allclose(inner, ref, atol=8e-3 * max(1.0, np.abs(ref).max()))
    if out_pad:
        assert (got[:, 0] == 0x7FC0).all() and (got[:, -1] == 0x7FC0).all() and (got[:, :, 0] == 0x7FC0).all() and (got[:, :, -1] == 0x7FC0).all()
    # argument checks: a tap that would leave the padded frame, a fused upsampling with the wrong output size
    bad = _abi.Conv16Desc(in_=be.ptr(XIN), W=be.ptr(WP), out=be.ptr(OUT), bias=None, residual=None, N=N, Hi=H, Wi=W, Cin=Cin, in_pad=0, Ho=H, Wo=W, Cout=Cout,
                          out_pad=out_pad, KS=3, stride=1, pad_top=0, pad_left=0, upsample=0, dtype=_abi.DT_BF16)
    assert be.lib.eegclip_conv16(bad, be.stream) < 0
    bad2 = _abi.Conv16Desc(in_=be.ptr(XIN), W=be.ptr(WP), out=be.ptr(OUT), bias=None, residual=None, N=N, Hi=H, Wi=W, Cin=Cin, in_pad=1, Ho=H, Wo=W, Cout=Cout,
                           out_pad=out_pad, KS=3, stride=1, pad_top=1, pad_left=1, upsample=1, dtype=_abi.DT_BF16)
    assert be.lib.eegclip_conv16(bad2, be.stream) < 0


def test_groupnorm16_softmax_and_sampling(be):
    rng = np.random.default_rng(9)
    N, C, H, W, G = 2, 64, 5, 6, 16
    x = bf(rng.standard_normal((N, C, H, W)) * 2 + 0.5)
    g, b = bf(1 + 0.2 * rng.standard_normal(C)), bf(0.2 * rng.standard_normal(C))
    X, GA, BE = be.dev(frame(x, 1)), be.dev(u16(g)), be.dev(u16(b))
    for silu in (0, 1):
        Y, S = be.dev(np.zeros((N, H + 2, W + 2, C), np.uint16)), be.dev(np.full(N * G * 2, np.nan, np.float64))
        ok(be.lib.eegclip_groupnorm16(be.ptr(X), N, H, W, C, 1, G, be.ptr(GA), be.ptr(BE), 1e-6, silu, be.ptr(Y), 1, be.ptr(S), _abi.DT_BF16, be.stream))
        be.sync()
        xg = x.astype(np.float64).reshape(N, G, -1)
        ref = ((xg - xg.mean(2, keepdims=True)) / np.sqrt(xg.var(2, keepdims=True) + 1e-6)).reshape(N, C, H, W) * g[None, :, None, None] + b[None, :, None, None]
        if silu:
            ref = ref / (1 + np.exp(-ref))
        got = be.host(Y)
        np.testing.assert_allclose(f32(got[:, 1:-1, 1:-1, :]).transpose(0, 3, 1, 2), ref, atol=2e-2)
        assert not got[:, 0].any() and not got[:, :, -1].any()
    s = bf(rng.standard_normal((5, 200)) * 3)
    SM = be.dev(u16(s))
    ok(be.lib.eegclip_softmax_rows16(be.ptr(SM), 5, 200, 200, 0.25, _abi.DT_BF16, be.stream))
    be.sync()
    e = np.exp(0.25 * s.astype(np.float64) - (0.25 * s.astype(np.float64)).max(1, keepdims=True))
    np.testing.assert_allclose(f32(be.host(SM)), e / e.sum(1, keepdims=True), atol=4e-3)
    mom, nz = bf(rng.standard_normal((7, 8))), bf(rng.standard_normal((7, 4)))
    mom[0, 4:] = 50.0                                                      # logvar is clamped to [-30, 20]
    M, NZ, Z = be.dev(u16(mom)), be.dev(u16(nz)), be.dev(np.zeros((7, 4), np.uint16))
    ok(be.lib.eegclip_vae_sample16(be.ptr(M), be.ptr(NZ), be.ptr(Z), 7, 4, _abi.DT_BF16, be.stream))
    be.sync()
    want = mom[:, :4].astype(np.float64) + np.exp(0.5 * np.clip(mom[:, 4:].astype(np.float64), -30, 20)) * nz
    np.testing.assert_allclose(f32(be.host(Z)), want, rtol=1e-2, atol=1e-2)
    ok(be.lib.eegclip_vae_sample16(be.ptr(M), None, be.ptr(Z), 7, 4, _abi.DT_BF16, be.stream))
    be.sync()
    np.testing.assert_array_equal(f32(be.host(Z)), mom[:, :4])
