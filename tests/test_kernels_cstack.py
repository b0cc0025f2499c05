"""csrc/cstack*.hip: the conv stack of Enc_eeg recomputed from the token rows (no y1 in HBM), against a plain torch-CPU fp64 reference of
Conv2d(1,40,(1,25)) -> AvgPool2d((1,51),(1,5)) -> BatchNorm2d -> ELU -> Conv2d(40,40,(H,1)) (Retrieval/ATMS_retrieval.py:102-106) and its backward,
on both backends (CPU lane emulator here, MI355X with -m gpu).  Products are split-bf16 (~2^-16 relative per term)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from backends import be, ok  # noqa: F401
from eeg_image_decode_amd import _abi

C, WD = 40, 36


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _problem(B, H, seed):
    rng = np.random.default_rng(seed)
    p = dict(x=rnd(rng, B, 64, 250), w25=rnd(rng, C, 25, scale=0.2), bias1=rnd(rng, C, scale=0.1), g1=(1 + 0.1 * rnd(rng, C)).astype(np.float32),
             b1=(0.1 * rnd(rng, C)).astype(np.float32), Ws=(rnd(rng, C, C, H) / np.sqrt(C * H)).astype(np.float32), bias2=(0.1 * rnd(rng, C)).astype(np.float32),
             dy2=rnd(rng, B, C, WD))
    return p


def _reference(p, B, H):
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=k in ("x", "w25", "g1", "b1", "Ws")) for k, v in p.items()}
    y1 = F.avg_pool2d(F.conv2d(t["x"][:, :H].unsqueeze(1), t["w25"].view(C, 1, 1, 25), t["bias1"]), (1, 51), (1, 5))
    z1 = F.elu(F.batch_norm(y1, None, None, t["g1"], t["b1"], True, 0.1, 1e-5))
    y2 = F.conv2d(z1, t["Ws"].view(C, C, H, 1), t["bias2"]).squeeze(2)
    return t, y1, y2


def _pack(be, p, H):
    n = int(be.lib.eegclip_cstack_packed_bytes(H))
    assert n > 0 and n % 1024 == 0
    PK = be.dev(np.full(n // 2, 0x7FC0, np.uint16))
    WS = be.dev(p["Ws"])
    ok(be.lib.eegclip_cstack_pack(be.ptr(WS), be.ptr(PK), H, be.stream))
    return PK


@pytest.mark.parametrize("B,H", [(2, 63), (3, 5), (2, 64), (5, 62)])
def test_cstack_forward(be, B, H):
    p = _problem(B, H, 11 * B + H)
    t, y1t, y2t = _reference(p, B, H)
    X, W25, BIAS1, G1, B1, BIAS2 = (be.dev(p[k]) for k in ("x", "w25", "bias1", "g1", "b1", "bias2"))
    PK = _pack(be, p, H)
    # BatchNorm1 partial rows: one [sum | sumsq] row per sample
    ROWS = be.dev(np.full((B, 80), np.nan, np.float64))
    ok(be.lib.eegclip_cstack_stats1(be.ptr(X), 64 * 250, 250, be.ptr(W25), be.ptr(BIAS1), be.ptr(ROWS), B, H, be.stream))
    rows = be.host(ROWS)
    y1 = y1t.detach().numpy()
    np.testing.assert_allclose(rows[:, :40], y1.sum((2, 3)), rtol=1e-5, atol=2e-3)       # (sums of 2268 values)
    np.testing.assert_allclose(rows[:, 40:], (y1 ** 2).sum((2, 3)), rtol=2e-4, atol=1e-5)
    count = float(B * H * WD)
    mean_ref, var_ref = y1.mean((0, 2, 3)), y1.var((0, 2, 3))
    for _ in range(2):                                      # (twice: nothing may depend on what a previous call left behind)
        MU, RS = be.dev(np.full(C, np.nan, np.float32)), be.dev(np.full(C, np.nan, np.float32))
        RM, RV = be.dev(np.full(C, 0.5, np.float32)), be.dev(np.full(C, 2.0, np.float32))
        NBT = be.dev(np.array([7], np.int64))
        Y2, ST2 = be.dev(np.full((B, C, WD), np.nan, np.float32)), be.dev(np.full((B, 80), np.nan, np.float64))
        d = _abi.CstackFwdDesc(B=B, H=H, x=be.ptr(X), xs_b=64 * 250, xs_h=250, w25=be.ptr(W25), bias1=be.ptr(BIAS1), stat1=be.ptr(ROWS), nstat1=B,
                               count1=count, eps=1e-5, momentum=0.1, gamma1=be.ptr(G1), beta1=be.ptr(B1), mean1=be.ptr(MU), rstd1=be.ptr(RS),
                               run_mean1=be.ptr(RM), run_var1=be.ptr(RV), nbt1=be.ptr(NBT), packed=be.ptr(PK), bias2=be.ptr(BIAS2), y2=be.ptr(Y2),
                               stat2=be.ptr(ST2))
        ok(be.lib.eegclip_cstack_fwd(ctypes.byref(d), be.stream))
        y2 = y2t.detach().numpy()
        np.testing.assert_allclose(be.host(Y2), y2, atol=1e-4)
        assert np.abs(be.host(Y2) - y2).mean() < 1.5e-5
        np.testing.assert_allclose(be.host(MU), mean_ref, atol=2e-6)
        np.testing.assert_allclose(be.host(RS), 1 / np.sqrt(var_ref + 1e-5), rtol=2e-4)
        np.testing.assert_allclose(be.host(RM), 0.9 * 0.5 + 0.1 * mean_ref, atol=1e-6)
        np.testing.assert_allclose(be.host(RV), 0.9 * 2.0 + 0.1 * var_ref * count / (count - 1), rtol=1e-5)
        assert int(be.host(NBT)[0]) == 8
        st2 = be.host(ST2)
        np.testing.assert_allclose(st2[:, :40], be.host(Y2).astype(np.float64).sum(2), atol=1e-5)
        np.testing.assert_allclose(st2[:, 40:], (be.host(Y2).astype(np.float64) ** 2).sum(2), rtol=1e-6)
    # eval mode: mean1 / rstd1 are inputs, no statistics are touched; all-reduced sums as one row
    MU, RS = be.dev(mean_ref.astype(np.float32)), be.dev((1 / np.sqrt(var_ref + 1e-5)).astype(np.float32))
    Y2 = be.dev(np.full((B, C, WD), np.nan, np.float32))
    d = _abi.CstackFwdDesc(B=B, H=H, x=be.ptr(X), xs_b=64 * 250, xs_h=250, w25=be.ptr(W25), bias1=be.ptr(BIAS1), stat1=None, nstat1=0, count1=0.0, eps=1e-5,
                           momentum=0.1, gamma1=be.ptr(G1), beta1=be.ptr(B1), mean1=be.ptr(MU), rstd1=be.ptr(RS), run_mean1=None, run_var1=None, nbt1=None,
                           packed=be.ptr(PK), bias2=be.ptr(BIAS2), y2=be.ptr(Y2), stat2=None)
    ok(be.lib.eegclip_cstack_fwd(ctypes.byref(d), be.stream))
    np.testing.assert_allclose(be.host(Y2), y2t.detach().numpy(), atol=1e-4)
    ONE = be.dev(rows.sum(0, keepdims=True))
    MU2, RS2 = be.dev(np.zeros(C, np.float32)), be.dev(np.zeros(C, np.float32))
    d.stat1, d.nstat1, d.count1, d.mean1, d.rstd1 = be.ptr(ONE), 1, count, be.ptr(MU2), be.ptr(RS2)
    ok(be.lib.eegclip_cstack_fwd(ctypes.byref(d), be.stream))
    np.testing.assert_allclose(be.host(Y2), y2t.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(be.host(MU2), mean_ref, atol=2e-6)
    # argument checks
    d.packed = None
    assert be.lib.eegclip_cstack_fwd(ctypes.byref(d), be.stream) < 0
    assert be.lib.eegclip_cstack_stats1(be.ptr(X), 64 * 250, 250, be.ptr(W25), be.ptr(BIAS1), be.ptr(ROWS), B, 65, be.stream) < 0
    assert be.lib.eegclip_cstack_pack(None, be.ptr(PK), H, be.stream) < 0
    assert int(be.lib.eegclip_cstack_packed_bytes(0)) == 0


@pytest.mark.parametrize("B,H", [(2, 63), (3, 5), (2, 64), (18, 7)])
def test_cstack_backward(be, B, H):
    """BatchNorm1-backward sums, the apply pass (token-row gradients, taps gradient, dgamma / dbeta) and the spatial-conv weight gradient, all recomputed
    from the token rows, against autograd of the fp64 reference"""
    p = _problem(B, H, 1000 + 11 * B + H)
    t, y1t, y2t = _reference(p, B, H)
    y2t.backward(t["dy2"])
    y1 = y1t.detach().numpy()
    mean_ref, var_ref = y1.mean((0, 2, 3)), y1.var((0, 2, 3))
    X, W25, BIAS1, G1, B1, DY2, WS = (be.dev(p[k]) for k in ("x", "w25", "bias1", "g1", "b1", "dy2", "Ws"))
    MU, RS = be.dev(mean_ref.astype(np.float32)), be.dev((1 / np.sqrt(var_ref + 1e-5)).astype(np.float32))
    nt = int(be.lib.eegclip_cstack_packed_t_bytes(H))
    assert nt == H * 9216
    PT = be.dev(np.full(nt // 2, 0x7FC0, np.uint16))
    ok(be.lib.eegclip_cstack_pack_t(be.ptr(WS), be.ptr(PT), H, be.stream))
    # both fragment sets in one launch == the two separate packs
    PK1 = _pack(be, p, H)
    PK2, PT2 = be.dev(np.full(int(be.lib.eegclip_cstack_packed_bytes(H)) // 2, 0x7FC0, np.uint16)), be.dev(np.full(nt // 2, 0x7FC0, np.uint16))
    ok(be.lib.eegclip_cstack_pack_all(be.ptr(WS), be.ptr(PK2), be.ptr(PT2), H, be.stream))
    assert np.array_equal(be.host(PK2), be.host(PK1)) and np.array_equal(be.host(PT2), be.host(PT))
    ROWS = be.dev(np.full((B, 80), np.nan, np.float64))
    DG, DB = be.dev(np.full(C, 0.25, np.float32)), be.dev(np.full(C, -0.25, np.float32))
    DX = be.dev(np.full((B, 64, 250), 7.0, np.float32))
    DWP = be.dev(np.full(int(be.lib.eegclip_cstack_bwd_workspace_floats(B)), np.nan, np.float32))
    DW25 = be.dev(np.ones((C, 25), np.float32))
    d = _abi.CstackBwdDesc(B=B, H=H, x=be.ptr(X), xs_b=64 * 250, xs_h=250, w25=be.ptr(W25), bias1=be.ptr(BIAS1), mean1=be.ptr(MU), rstd1=be.ptr(RS),
                           gamma1=be.ptr(G1), beta1=be.ptr(B1), packed_t=be.ptr(PT), dy2=be.ptr(DY2), rows_out=be.ptr(ROWS), stat=be.ptr(ROWS), nstat=B,
                           count=float(B * H * WD), stat_local=None, nstat_local=0, dgamma=be.ptr(DG), dbeta=be.ptr(DB), dx=be.ptr(DX),
                           dw_partials=be.ptr(DWP), dw25=be.ptr(DW25))
    ok(be.lib.eegclip_cstack_bwd_stats(ctypes.byref(d), be.stream))
    rows = be.host(ROWS)
    gb, gg = t["b1"].grad.numpy(), t["g1"].grad.numpy()
    np.testing.assert_allclose(rows[:, :40].sum(0), gb, atol=2e-4 * max(1.0, np.abs(gb).max()))
    np.testing.assert_allclose(rows[:, 40:].sum(0), gg, atol=2e-4 * max(1.0, np.abs(gg).max()))
    ok(be.lib.eegclip_cstack_bwd_apply(ctypes.byref(d), be.stream))
    dx, gx = be.host(DX), t["x"].grad.numpy()
    np.testing.assert_array_equal(dx[:, H:], 7.0)                    # token rows past H are not touched
    np.testing.assert_allclose(dx[:, :H], gx[:, :H], atol=3e-5 * max(1.0, float(np.abs(gx).max())))
    gw = t["w25"].grad.numpy()
    np.testing.assert_allclose(be.host(DW25) - 1.0, gw, atol=2e-4 * max(1.0, np.abs(gw).max()))
    np.testing.assert_allclose(be.host(DB) + 0.25, gb, atol=2e-4 * max(1.0, np.abs(gb).max()))
    np.testing.assert_allclose(be.host(DG) - 0.25, gg, atol=2e-4 * max(1.0, np.abs(gg).max()))
    # data-parallel form: all-reduced sums as ONE row, dgamma / dbeta from the rank's own rows; eval-mode BatchNorm: a zero row drops the batch terms
    ONE = be.dev(rows.sum(0, keepdims=True))
    DG2, DB2, DX2 = be.zeros(C), be.zeros(C), be.dev(np.full((B, 64, 250), 7.0, np.float32))
    d.stat, d.nstat, d.stat_local, d.nstat_local, d.dgamma, d.dbeta, d.dx = be.ptr(ONE), 1, be.ptr(ROWS), B, be.ptr(DG2), be.ptr(DB2), be.ptr(DX2)
    d.dw25 = None                                                    # ... and the tap gradient left as rows for a launch of its own (second stream in the plans)
    ok(be.lib.eegclip_cstack_bwd_apply(ctypes.byref(d), be.stream))
    DW25b = be.dev(np.ones((C, 25), np.float32))
    ok(be.lib.eegclip_cstack_bwd_taps_reduce(be.ptr(DWP), B, be.ptr(DW25b), be.stream))
    np.testing.assert_array_equal(be.host(DW25b), be.host(DW25))
    assert be.lib.eegclip_cstack_bwd_taps_reduce(None, B, be.ptr(DW25b), be.stream) < 0 and be.lib.eegclip_cstack_bwd_taps_reduce(be.ptr(DWP), 0, be.ptr(DW25b), be.stream) < 0
    d.dw25 = be.ptr(DW25)
    np.testing.assert_allclose(be.host(DX2), dx, atol=1e-6)
    np.testing.assert_allclose(be.host(DB2), gb, atol=2e-4 * max(1.0, np.abs(gb).max()))
    np.testing.assert_allclose(be.host(DG2), gg, atol=2e-4 * max(1.0, np.abs(gg).max()))
    # spatial-conv weight gradient
    DWS = be.dev(np.ones((C, C, H), np.float32))
    WSP = be.dev(np.full(int(be.lib.eegclip_cstack_bwd_w2_workspace_floats(B, H)), np.nan, np.float32))
    ok(be.lib.eegclip_cstack_bwd_w2(be.ptr(X), 64 * 250, 250, be.ptr(W25), be.ptr(BIAS1), be.ptr(MU), be.ptr(RS), be.ptr(G1), be.ptr(B1), be.ptr(DY2),
                                    be.ptr(DWS), be.ptr(WSP), B, H, be.stream))
    gws = t["Ws"].grad.numpy()
    np.testing.assert_allclose(be.host(DWS) - 1.0, gws, atol=1e-4 * max(1.0, np.abs(gws).max()))
    # argument checks
    d.packed_t = None
    assert be.lib.eegclip_cstack_bwd_stats(ctypes.byref(d), be.stream) < 0
    assert be.lib.eegclip_cstack_bwd_apply(ctypes.byref(d), be.stream) < 0
    assert be.lib.eegclip_cstack_bwd_w2(be.ptr(X), 64 * 250, 250, be.ptr(W25), be.ptr(BIAS1), be.ptr(MU), be.ptr(RS), be.ptr(G1), be.ptr(B1), be.ptr(DY2),
                                        be.ptr(DWS), None, B, H, be.stream) < 0
