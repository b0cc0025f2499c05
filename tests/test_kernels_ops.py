"""Per-kernel parity: every C-ABI op vs a plain torch-CPU fp64 reference of the same op, on both backends
(CPU lane emulator here, MI355X with -m gpu).  Tolerances are fp32 round-off (1e-5 .. 1e-4 relative)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from backends import be, ok  # noqa: F401
from eeg_image_decode_amd import _abi
from philox_np import keep_mask

SEED = 0x5EEDC0FFEE


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("rows,cols", [(5, 250), (64, 1024), (130, 64), (3, 7), (16391, 6)])
def test_layernorm_fwd_bwd(be, rows, cols):
    rng = np.random.default_rng(rows * 7 + cols)
    x, g, b, dy = rnd(rng, rows, cols), 1 + 0.1 * rnd(rng, cols), 0.1 * rnd(rng, cols), rnd(rng, rows, cols)
    X, G, Bt, DY = be.dev(x), be.dev(g), be.dev(b), be.dev(dy)
    Y, MU, RS = be.zeros((rows, cols)), be.zeros(rows), be.zeros(rows)
    ok(be.lib.eegclip_layernorm_fwd(be.ptr(X), be.ptr(G), be.ptr(Bt), be.ptr(Y), be.ptr(MU), be.ptr(RS), rows, cols, 1e-5, be.stream))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(g, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    yt = F.layer_norm(xt, (cols,), gt, bt, 1e-5)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    np.testing.assert_allclose(be.host(Y), yt.detach().numpy(), atol=2e-5)
    dx0 = rnd(rng, rows, cols)
    DX, DG, DB = be.dev(dx0), be.zeros(cols), be.zeros(cols)
    DXD = be.zeros((rows, cols))
    ok(be.lib.eegclip_layernorm_bwd(be.ptr(DY), be.ptr(X), be.ptr(G), be.ptr(MU), be.ptr(RS), be.ptr(DX), be.ptr(DG), be.ptr(DB),
                                    rows, cols, 1, be.ptr(DXD), 0.25, SEED, 4, be.stream))
    np.testing.assert_allclose(be.host(DX), dx0 + xt.grad.numpy(), atol=5e-5)
    keep = keep_mask(SEED, 4, rows * cols, 0.25).reshape(rows, cols)          # second output: dx pushed back through a dropout
    np.testing.assert_allclose(be.host(DXD), (dx0 + xt.grad.numpy()) * keep / 0.75, atol=1e-4)
    np.testing.assert_allclose(be.host(DG), gt.grad.numpy(), atol=1e-4 * max(1, rows ** 0.5))
    np.testing.assert_allclose(be.host(DB), bt.grad.numpy(), atol=1e-4 * max(1, rows ** 0.5))
    # the two halves on their own (the encoder's backward runs the parameter half on its second stream)
    DX2, DG2, DB2 = be.dev(dx0), be.zeros(cols), be.zeros(cols)
    ok(be.lib.eegclip_layernorm_bwd(be.ptr(DY), be.ptr(X), be.ptr(G), be.ptr(MU), be.ptr(RS), be.ptr(DX2), None, None, rows, cols, 1, None, 0.0, 0, 0,
                                    be.stream))
    ok(be.lib.eegclip_layernorm_bwd(be.ptr(DY), be.ptr(X), None, be.ptr(MU), be.ptr(RS), None, be.ptr(DG2), be.ptr(DB2), rows, cols, 0, None, 0.0, 0, 0,
                                    be.stream))
    assert np.array_equal(be.host(DX2), be.host(DX))
    np.testing.assert_allclose(be.host(DG2), be.host(DG), atol=1e-4 * max(1, rows ** 0.5))         # atomics: summation order varies
    np.testing.assert_allclose(be.host(DB2), be.host(DB), atol=1e-4 * max(1, rows ** 0.5))
    assert be.lib.eegclip_layernorm_bwd(be.ptr(DY), be.ptr(X), be.ptr(G), be.ptr(MU), be.ptr(RS), None, None, None, rows, cols, 0, None, 0.0, 0, 0,
                                        be.stream) < 0
    # both halves in one pass: dx, dropout'(dx) and the parameter gradients (partial rows + column reduction)
    nwf = int(be.lib.eegclip_layernorm_bwd_full_workspace_floats(rows, cols))
    WSF, DX4, DXD4 = be.dev(np.full(nwf, np.nan, np.float32)), be.dev(dx0), be.zeros((rows, cols))
    DG4, DB4 = be.dev(np.full(cols, 0.5, np.float32)), be.dev(np.full(cols, -0.5, np.float32))
    ok(be.lib.eegclip_layernorm_bwd_full(be.ptr(DY), be.ptr(X), be.ptr(G), be.ptr(MU), be.ptr(RS), be.ptr(DX4), be.ptr(DG4), be.ptr(DB4), rows, cols, 1,
                                         be.ptr(DXD4), 0.25, SEED, 4, be.ptr(WSF), be.stream))
    if cols == 1024:      # (rows of 1024: eegclip_layernorm_bwd's input-gradient half runs one workgroup per row, the one-pass form one wave per row: other summation order)
        np.testing.assert_allclose(be.host(DX4), be.host(DX), atol=2e-6)
        np.testing.assert_allclose(be.host(DXD4), be.host(DXD), atol=3e-6)
    else:
        assert np.array_equal(be.host(DX4), be.host(DX)) and np.array_equal(be.host(DXD4), be.host(DXD))
    np.testing.assert_allclose(be.host(DG4) - 0.5, gt.grad.numpy(), atol=1e-4 * max(1, rows ** 0.5))
    np.testing.assert_allclose(be.host(DB4) + 0.5, bt.grad.numpy(), atol=1e-4 * max(1, rows ** 0.5))
    # the parameter half without atomics on the inputs' scale: per-workgroup partial rows in a workspace + a column reduction; accumulates
    nws = int(be.lib.eegclip_layernorm_bwd_params_workspace_floats(rows, cols))
    assert nws == (rows + 15) // 16 * 2 * cols
    WS, DG3, DB3 = be.dev(np.full(nws, np.nan, np.float32)), be.dev(np.full(cols, 2.0, np.float32)), be.dev(np.full(cols, -1.0, np.float32))
    ok(be.lib.eegclip_layernorm_bwd_params(be.ptr(DY), be.ptr(X), be.ptr(MU), be.ptr(RS), be.ptr(DG3), be.ptr(DB3), rows, cols, be.ptr(WS), be.stream))
    np.testing.assert_allclose(be.host(DG3) - 2.0, gt.grad.numpy(), atol=1e-4 * max(1, rows ** 0.5))
    np.testing.assert_allclose(be.host(DB3) + 1.0, bt.grad.numpy(), atol=1e-4 * max(1, rows ** 0.5))


@pytest.mark.parametrize("rows,cols,p,double", [(5, 250, 0.25, True), (130, 250, 0.25, False), (7, 64, 0.0, True), (3, 1024, 0.5, False), (4, 7, 0.25, True)])
def test_residual_layernorm_fwd(be, rows, cols, p, double):
    """v = resid + dropout(x) -> x_out (in place over x); y = LN(v); optionally y2 = LN2(y): the encoder's post-LN sublayer tail with the
    dropout mask of the GEMM epilogue it replaces (Philox(seed, site, row*cols + c))"""
    rng = np.random.default_rng(rows + cols)
    x, r = rnd(rng, rows, cols), rnd(rng, rows, cols)
    g1, b1, g2, b2 = 1 + 0.1 * rnd(rng, cols), 0.1 * rnd(rng, cols), 1 + 0.1 * rnd(rng, cols), 0.1 * rnd(rng, cols)
    X, Rz, G1, B1, G2, B2 = be.dev(x), be.dev(r), be.dev(g1), be.dev(b1), be.dev(g2), be.dev(b2)
    Y, Y2, MU, RS, MU2, RS2 = be.zeros((rows, cols)), be.zeros((rows, cols)), be.zeros(rows), be.zeros(rows), be.zeros(rows), be.zeros(rows)
    ok(be.lib.eegclip_residual_layernorm_fwd(be.ptr(X), be.ptr(Rz), be.ptr(X), p, SEED, 6, be.ptr(G1), be.ptr(B1), be.ptr(Y), be.ptr(MU), be.ptr(RS),
                                             be.ptr(G2) if double else None, be.ptr(B2) if double else None, be.ptr(Y2) if double else None,
                                             be.ptr(MU2) if double else None, be.ptr(RS2) if double else None, rows, cols, 1e-5, be.stream))
    keep = keep_mask(SEED, 6, rows * cols, p).reshape(rows, cols) if p > 0 else np.ones((rows, cols), bool)
    ks = np.float32(1.0) / (np.float32(1.0) - np.float32(p))               # the kernels multiply by the fp32 reciprocal
    v = np.where(keep, x * ks, np.float32(0)).astype(np.float32) + r
    assert np.array_equal(be.host(X), v)                                   # same arithmetic order as the GEMM epilogue: bit exact
    vt = torch.tensor(v, dtype=torch.float64)
    y = F.layer_norm(vt, (cols,), torch.tensor(g1, dtype=torch.float64), torch.tensor(b1, dtype=torch.float64), 1e-5)
    np.testing.assert_allclose(be.host(Y), y.numpy(), atol=3e-5)
    np.testing.assert_allclose(be.host(MU), v.astype(np.float64).mean(1), atol=1e-5)
    np.testing.assert_allclose(be.host(RS), 1 / np.sqrt(v.astype(np.float64).var(1) + 1e-5), rtol=1e-4)
    if double:
        y2 = F.layer_norm(y, (cols,), torch.tensor(g2, dtype=torch.float64), torch.tensor(b2, dtype=torch.float64), 1e-5)
        np.testing.assert_allclose(be.host(Y2), y2.numpy(), atol=5e-5)
        np.testing.assert_allclose(be.host(MU2), y.numpy().mean(1), atol=2e-5)
    # plain LayerNorm form (no residual) and argument errors
    Y3 = be.zeros((rows, cols))
    ok(be.lib.eegclip_residual_layernorm_fwd(be.ptr(Rz), None, None, 0.0, 0, 0, be.ptr(G1), be.ptr(B1), be.ptr(Y3), None, None, None, None, None, None, None,
                                             rows, cols, 1e-5, be.stream))
    y3 = F.layer_norm(torch.tensor(r, dtype=torch.float64), (cols,), torch.tensor(g1, dtype=torch.float64), torch.tensor(b1, dtype=torch.float64), 1e-5)
    np.testing.assert_allclose(be.host(Y3), y3.numpy(), atol=3e-5)
    assert be.lib.eegclip_residual_layernorm_fwd(be.ptr(Rz), None, None, 0.25, 0, 0, be.ptr(G1), be.ptr(B1), be.ptr(Y3), None, None, None, None, None,
                                                 None, None, rows, cols, 1e-5, be.stream) < 0
    # the same pass leaving y as bf16 hi | lo planes as well (the InfoNCE kernels' operand form): same y, planes = the split eegclip_split_rows makes of it
    YH, YL = be.dev(np.zeros((rows, cols), np.uint16)), be.dev(np.zeros((rows, cols), np.uint16))
    Y4 = be.zeros((rows, cols))
    rc = be.lib.eegclip_residual_layernorm_fwd_planes(be.ptr(Rz), None, None, 0.0, 0, 0, be.ptr(G1), be.ptr(B1), be.ptr(Y4), None, None, None, None, None, None,
                                                      None, rows, cols, 1e-5, be.ptr(YH), be.ptr(YL), be.stream)
    if cols % 4:
        assert rc < 0
    else:
        ok(rc)
        y4 = be.host(Y4)
        np.testing.assert_array_equal(y4, be.host(Y3))
        hi = (be.host(YH).astype(np.uint32) << 16).view(np.float32)
        lo = (be.host(YL).astype(np.uint32) << 16).view(np.float32)
        np.testing.assert_array_equal(hi, _bf16_round(y4))
        np.testing.assert_array_equal(lo, _bf16_round((y4 - _bf16_round(y4)).astype(np.float32)))


@pytest.mark.parametrize("outer,C,inner,p", [(6, 40, 63 * 36, 0.0), (9, 40, 36, 0.5), (2, 3, 5, 0.0)])
def test_batchnorm_elu_train_fwd_bwd(be, outer, C, inner, p):
    rng = np.random.default_rng(outer + C + inner)
    x, g, b = rnd(rng, outer, C, inner) * 1.5 + 0.3, 1 + 0.1 * rnd(rng, C), 0.1 * rnd(rng, C)
    dz = rnd(rng, outer, C, inner)
    rm0, rv0 = 0.1 * rnd(rng, C), (1 + 0.2 * rng.random(C)).astype(np.float32)
    X, G, Bt, DZ = be.dev(x), be.dev(g), be.dev(b), be.dev(dz)
    SUMS = be.zeros(2 * C, np.float64)
    MU, RS, RM, RV = be.zeros(C), be.zeros(C), be.dev(rm0), be.dev(rv0)
    NBT = be.dev(np.array([7], np.int64))
    n = outer * inner
    ok(be.lib.eegclip_bn_stats(be.ptr(X), outer, C, inner, be.ptr(SUMS), be.stream))
    ok(be.lib.eegclip_bn_finalize(be.ptr(SUMS), float(n), 1e-5, 0.1, C, be.ptr(MU), be.ptr(RS), be.ptr(RM), be.ptr(RV), 1, be.ptr(NBT), be.stream))
    assert int(be.host(NBT)[0]) == 8
    Y = be.zeros((outer, C, inner))
    ok(be.lib.eegclip_bn_elu_fwd(be.ptr(X), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(Bt), be.ptr(Y), outer, C, inner, p, SEED, 5, be.stream))
    keep = keep_mask(SEED, 5, outer * C * inner, p).reshape(outer, C, inner) if p > 0 else np.ones((outer, C, inner), bool)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(g, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    rm, rv = torch.tensor(rm0, dtype=torch.float64), torch.tensor(rv0, dtype=torch.float64)
    yt = F.elu(F.batch_norm(xt, rm, rv, gt, bt, True, 0.1, 1e-5)) * torch.tensor(keep) / (1 - p)
    yt.backward(torch.tensor(dz, dtype=torch.float64))
    np.testing.assert_allclose(be.host(Y), yt.detach().numpy(), atol=3e-5)
    np.testing.assert_allclose(be.host(RM), rm.numpy(), atol=1e-6)
    np.testing.assert_allclose(be.host(RV), rv.numpy(), atol=1e-5)
    S2, DX, DG, DB = be.zeros(2 * C, np.float64), be.zeros((outer, C, inner)), be.zeros(C), be.zeros(C)
    ok(be.lib.eegclip_bn_elu_bwd(be.ptr(DZ), be.ptr(X), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(Bt), be.ptr(S2), be.ptr(DX), be.ptr(DG),
                                 be.ptr(DB), outer, C, inner, p, SEED, 5, be.stream))
    np.testing.assert_allclose(be.host(DX), xt.grad.numpy(), atol=5e-5)
    np.testing.assert_allclose(be.host(DG), gt.grad.numpy(), atol=2e-4 * max(1, n ** 0.5 / 10))
    np.testing.assert_allclose(be.host(DB), bt.grad.numpy(), atol=2e-4 * max(1, n ** 0.5 / 10))
    # eval mode: statistics come from the running buffers
    ok(be.lib.eegclip_bn_finalize(None, 0.0, 1e-5, 0.1, C, be.ptr(MU), be.ptr(RS), be.ptr(RM), be.ptr(RV), 0, be.ptr(NBT), be.stream))
    assert int(be.host(NBT)[0]) == 8             # eval leaves the step counter alone
    ok(be.lib.eegclip_bn_elu_fwd(be.ptr(X), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(Bt), be.ptr(Y), outer, C, inner, 0.0, 0, 0, be.stream))
    ye = F.elu(F.batch_norm(xt.detach(), rm, rv, gt.detach(), bt.detach(), False, 0.1, 1e-5))
    np.testing.assert_allclose(be.host(Y), ye.numpy(), atol=3e-5)


@pytest.mark.parametrize("use_ids,p", [(True, 0.0), (False, 0.25), (True, 0.25)])
def test_embed_finish_fwd_bwd(be, use_ids, p):
    rng = np.random.default_rng(3)
    B, L, D = 5, 64, 250
    h = rnd(rng, B, L, D)
    tokens = rnd(rng, 10, D)
    ids = np.array([1, 9, 3, 3, 0], np.int64)
    H, TK, IDS = be.dev(h), be.dev(tokens), be.dev(ids) if use_ids else None
    ok(be.lib.eegclip_embed_finish(be.ptr(H), be.ptr(TK), be.ptr(IDS), B, L, D, p, SEED, 0, be.stream))
    ref = h.copy()
    ref[:, 0] = tokens[ids] if use_ids else tokens[0]
    keep = keep_mask(SEED, 0, B * L * D, p).reshape(B, L, D) if p > 0 else np.ones((B, L, D), bool)
    np.testing.assert_allclose(be.host(H), ref * keep / (1 - p), rtol=1e-6)
    dh = rnd(rng, B, L, D)
    DH, DT = be.dev(dh), be.zeros((10, D))
    ok(be.lib.eegclip_embed_finish_bwd(be.ptr(DH), be.ptr(DT), be.ptr(IDS), B, L, D, p, SEED, 0, be.stream))
    dref = dh * keep / (1 - p)
    np.testing.assert_allclose(be.host(DH), dref, rtol=1e-6)
    dt = np.zeros((10, D))
    for b in range(B):
        dt[ids[b] if use_ids else 0] += dref[b, 0]
    np.testing.assert_allclose(be.host(DT), dt, atol=1e-5)


def test_elementwise_helpers(be):
    rng = np.random.default_rng(4)
    n = 10007
    x, y, pre = rnd(rng, n), rnd(rng, n), rnd(rng, n)
    X, Y = be.dev(x), be.dev(y)
    ok(be.lib.eegclip_axpby(be.ptr(X), be.ptr(Y), n, 2.0, -0.5, be.stream))
    np.testing.assert_allclose(be.host(Y), 2 * x - 0.5 * y, rtol=1e-6, atol=1e-6)
    Y2 = be.dev(y)
    ok(be.lib.eegclip_dropout_scale(be.ptr(Y2), n, 0.25, SEED, 2, be.stream))
    keep = keep_mask(SEED, 2, n, 0.25)
    np.testing.assert_allclose(be.host(Y2), y * keep / 0.75, rtol=1e-6)
    PRE, DX = be.dev(pre), be.dev(y)
    ok(be.lib.eegclip_gelu_bwd(be.ptr(X), be.ptr(PRE), be.ptr(DX), n, 1, 0.25, SEED, 2, be.stream))
    pt = torch.tensor(pre, dtype=torch.float64, requires_grad=True)
    (F.gelu(pt) * torch.tensor(keep) / 0.75).backward(torch.tensor(x, dtype=torch.float64))
    np.testing.assert_allclose(be.host(DX), y + pt.grad.numpy(), atol=1e-5)
    SS = be.zeros(1, np.float64)
    ok(be.lib.eegclip_sumsq(be.ptr(X), n, be.ptr(SS), be.stream))
    assert abs(be.host(SS)[0] - (x.astype(np.float64) ** 2).sum()) < 1e-6 * n
    SC = be.zeros(1)
    ok(be.lib.eegclip_clip_scale(be.ptr(SS), 1.0, be.ptr(SC), be.stream))
    assert abs(be.host(SC)[0] - min(1.0, 1.0 / (np.sqrt((x.astype(np.float64) ** 2).sum()) + 1e-6))) < 1e-7


@pytest.mark.parametrize("n,row,ds,ss", [(7, 15750, 16000, 15750), (1, 2, 2, 2), (33, 250, 250, 1000)])
def test_gather_scatter_rows_bit_exact(be, n, row, ds, ss):
    """sample-block gather (dst[j] = src[idx[j]]) and scatter (dst[idx[j]] = src[j]) with independent strides: pure data movement, bit exact"""
    rng = np.random.default_rng(12)
    perm = rng.permutation(n).astype(np.int32)
    src = rnd(rng, n * ss)
    dst0 = rnd(rng, n * ds)
    IDX, SRC = be.dev(perm), be.dev(src)
    for scatter in (0, 1):
        DST = be.dev(dst0)
        ok(be.lib.eegclip_gather_rows(be.ptr(DST), ds, be.ptr(SRC), ss, be.ptr(IDX), n, row, scatter, be.stream))
        want = dst0.copy()
        for j in range(n):
            a, b_ = (perm[j], j) if scatter else (j, perm[j])
            want[a * ds:a * ds + row] = src[b_ * ss:b_ * ss + row]
        assert np.array_equal(be.host(DST), want)                     # elements between blocks untouched
    assert be.lib.eegclip_gather_rows(be.ptr(DST), ds, be.ptr(DST), ss, be.ptr(IDX), n, 3, 0, be.stream) < 0       # odd row length


@pytest.mark.parametrize("n,reps,C,T,win", [(7, 4, 3, 10, (2, 9)), (5, 3, 2, 6, None), (33, 4, 63, 26, None), (4, 5, 3, 8, (0, 8))])
def test_stage_eeg_cast_window_and_mean(be, n, reps, C, T, win):
    """dataset staging (eegdatasets_leaveone.py:157,220,293-306): float64 trials -> float32, time window, optional mean over repetitions taken
    AFTER the float32 cast; tidx = NULL keeps every sample (flat vectorised cast)"""
    rng = np.random.default_rng(n + T)
    src = rng.standard_normal((n, reps, C, T))
    cols = np.arange(T) if win is None else np.arange(win[0], win[1])
    SRC = be.dev(src)
    TI = None if win is None else be.dev(cols.astype(np.int32))
    Tw = len(cols)
    DST = be.zeros((n * reps, C, Tw))
    ok(be.lib.eegclip_stage_eeg(be.ptr(SRC), be.ptr(DST), n, reps, C, T, be.ptr(TI), Tw, 0, be.stream))
    assert np.array_equal(be.host(DST), src.astype(np.float32)[..., cols].reshape(n * reps, C, Tw))
    if win is None:
        TI = be.dev(cols.astype(np.int32))                           # the mean path always takes an index list or NULL; exercise both
    for ti in (TI, None) if win is None else (TI,):
        DM = be.zeros((n, C, Tw))
        ok(be.lib.eegclip_stage_eeg(be.ptr(SRC), be.ptr(DM), n, reps, C, T, be.ptr(ti), Tw, 1, be.stream))
        f = src.astype(np.float32)[..., cols]
        acc = np.zeros((n, C, Tw), np.float32)
        for r in range(reps):
            acc += f[:, r]
        assert np.array_equal(be.host(DM), acc / np.float32(reps))          # float32 accumulation in repetition order, bit for bit
    assert be.lib.eegclip_stage_eeg(be.ptr(SRC), be.ptr(DST), n, reps, C, T, None, Tw - 1 if Tw > 1 else 0, 0, be.stream) < 0


@pytest.mark.parametrize("outer,mid,inner", [(1000, 250, 1), (7, 40, 36), (3, 1024, 1), (1, 5, 1)])
def test_reduce_mid(be, outer, mid, inner):
    rng = np.random.default_rng(outer)
    x = rnd(rng, outer, mid, inner)
    o0 = rnd(rng, mid)
    X, O = be.dev(x), be.dev(o0)
    ok(be.lib.eegclip_reduce_mid(be.ptr(X), outer, mid, inner, be.ptr(O), be.stream))
    np.testing.assert_allclose(be.host(O), o0 + x.astype(np.float64).sum(axis=(0, 2)), atol=1e-4 * max(1, (outer * inner) ** 0.5 / 8))


def test_adamw_matches_torch(be):
    rng = np.random.default_rng(6)
    n = 5000
    p0 = rnd(rng, n)
    P, M, V = be.dev(p0), be.zeros(n), be.zeros(n)
    pt = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.AdamW([pt], lr=3e-4)
    for step in range(1, 4):
        g = rnd(rng, n) * (1e-3 if step == 2 else 1.0)
        G = be.dev(g)
        ok(be.lib.eegclip_adamw_step(be.ptr(P), be.ptr(G), be.ptr(M), be.ptr(V), n, 3e-4, 0.9, 0.999, 1e-8, 0.01, step, 1.0, None, be.stream))
        pt.grad = torch.tensor(g)
        opt.step()
    np.testing.assert_allclose(be.host(P), pt.detach().numpy(), atol=2e-7, rtol=1e-6)


def test_adamw_step_zero_grad_is_the_same_update_and_clears_the_gradient(be):
    rng = np.random.default_rng(16)
    n = 4099
    p0, g, m0, v0 = rnd(rng, n), rnd(rng, n), rnd(rng, n) * 0.1, np.abs(rnd(rng, n)) * 0.1
    res = []
    for name in ("eegclip_adamw_step", "eegclip_adamw_step_zero_grad"):
        P, G, M, V = be.dev(p0), be.dev(g), be.dev(m0), be.dev(v0)
        ok(getattr(be.lib, name)(be.ptr(P), be.ptr(G), be.ptr(M), be.ptr(V), n, 3e-4, 0.9, 0.999, 1e-8, 0.01, 3, 0.5, None, be.stream))
        res.append([be.host(t) for t in (P, M, V, G)])
    for a, b in zip(res[0][:3], res[1][:3]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[0][3], g)
    np.testing.assert_array_equal(res[1][3], np.zeros(n, np.float32))


def _attn_ref(qkv, B, H, E, scale, keep, p):
    L = 64
    q, k, v = [t.view(B, L, H, E).permute(0, 2, 1, 3) for t in qkv.view(B * L, 3, H * E).unbind(1)]
    A = torch.softmax(q @ k.transpose(-1, -2) * scale, -1)
    A = A * keep / (1 - p)
    return (A @ v).permute(0, 2, 1, 3).reshape(B * L, H * E)


@pytest.mark.parametrize("B,H,E,p", [(2, 4, 62, 0.0), (3, 4, 62, 0.25), (1, 2, 64, 0.0), (1, 1, 5, 0.0)])
def test_attention_fwd_bwd(be, B, H, E, p):
    rng = np.random.default_rng(B * 100 + H * 10 + E)
    L, ld = 64, 3 * H * E
    qkv = rnd(rng, B * L, ld, scale=0.8)
    dctx = rnd(rng, B * L, H * E)
    scale = 1.0 / math.sqrt(E)
    QKV, DCTX = be.dev(qkv), be.dev(dctx)
    CTX, DQKV = be.zeros((B * L, H * E)), be.zeros((B * L, ld))
    ok(be.lib.eegclip_attention_fwd(be.ptr(QKV), be.ptr(CTX), B, L, H, E, ld, scale, p, SEED, 1, be.stream))
    ok(be.lib.eegclip_attention_bwd(be.ptr(QKV), be.ptr(DCTX), be.ptr(DQKV), B, L, H, E, ld, scale, p, SEED, 1, be.stream))
    keep = torch.tensor(keep_mask(SEED, 1, B * H * L * L, p).reshape(B, H, L, L)) if p > 0 else torch.ones(B, H, L, L, dtype=torch.bool)
    qt = torch.tensor(qkv, dtype=torch.float64, requires_grad=True)
    out = _attn_ref(qt, B, H, E, scale, keep, p)
    out.backward(torch.tensor(dctx, dtype=torch.float64))
    np.testing.assert_allclose(be.host(CTX), out.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(be.host(DQKV), qt.grad.numpy(), atol=5e-5)
    if E % 2 == 0:          # the split-bf16 backward (csrc/attention_x3.hip: LDS transpose reads): same masks, ~2^-16 relative per product term
        DQ3 = be.dev(np.full((B * L, ld), np.nan, np.float32))
        ok(be.lib.eegclip_attention_bwd_x3(be.ptr(QKV), be.ptr(DCTX), be.ptr(DQ3), 0, B, L, H, E, ld, scale, p, SEED, 1, be.stream))
        np.testing.assert_allclose(be.host(DQ3), qt.grad.numpy(), atol=2e-4)
        assert np.abs(be.host(DQ3) - qt.grad.numpy()).mean() < 1e-5
        # ... and the same gradients as token planes (what the fused backward + the q | k | v weight gradient read): hi + lo = the fp32 value to 2^-17,
        # channel 64 head + e, the channels past E exact zeros, channels of absent heads untouched
        DQP = be.dev(np.full((3, B, 2, L, 256), 0x7FC0, np.uint16))
        ok(be.lib.eegclip_attention_bwd_x3(be.ptr(QKV), be.ptr(DCTX), be.ptr(DQP), 1, B, L, H, E, ld, scale, p, SEED, 1, be.stream))
        pl = (be.host(DQP).astype(np.uint32) << 16).view(np.float32)
        val = (pl[:, :, 0] + pl[:, :, 1])[..., :64 * H].reshape(3, B, L, H, 64)
        np.testing.assert_array_equal(val[..., E:], 0.0)
        got = val[..., :E].transpose(1, 2, 0, 3, 4).reshape(B * L, ld)
        ref3 = be.host(DQ3)
        np.testing.assert_allclose(got, ref3, atol=2e-5 * max(1.0, float(np.abs(ref3).max())))
        assert np.all(be.host(DQP)[..., 64 * H:] == 0x7FC0)
    else:
        assert be.lib.eegclip_attention_bwd_x3(be.ptr(QKV), be.ptr(DCTX), be.ptr(DQKV), 0, B, L, H, E, ld, scale, p, SEED, 1, be.stream) < 0


@pytest.mark.parametrize("B,H", [(2, 63), (3, 5), (5, 63)])
def test_tsconv_fwd_bwd(be, B, H):
    rng = np.random.default_rng(B + H)
    w25, bias = rnd(rng, 40, 25, scale=0.2), rnd(rng, 40, scale=0.1)
    xfull = rnd(rng, B, 64, 250)                       # encoder output; rows h < H are convolved in place
    W25, BIAS, X = be.dev(w25), be.dev(bias), be.dev(xfull)
    Y, SUMS = be.zeros((B, 40, H, 36)), be.zeros(80, np.float64)
    ok(be.lib.eegclip_tsconv_fwd(be.ptr(X), 64 * 250, 250, be.ptr(W25), be.ptr(BIAS), be.ptr(Y), B, H, 250, 40, be.ptr(SUMS), None, be.stream))
    SUMS_W = be.zeros(80, np.float64)           # the same sums through per-workgroup partial rows + a column reduction
    WSF = be.dev(np.full(int(be.lib.eegclip_tsconv_fwd_workspace_floats(B, H)) // 2, np.nan, np.float64))
    ok(be.lib.eegclip_tsconv_fwd(be.ptr(X), 64 * 250, 250, be.ptr(W25), be.ptr(BIAS), be.ptr(Y), B, H, 250, 40, be.ptr(SUMS_W), be.ptr(WSF), be.stream))
    np.testing.assert_allclose(be.host(SUMS_W), be.host(SUMS), rtol=1e-9, atol=1e-9)
    xt = torch.tensor(xfull, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w25, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(bias, dtype=torch.float64)
    yt = F.avg_pool2d(F.conv2d(xt[:, :H].unsqueeze(1), wt.view(40, 1, 1, 25), bt), (1, 51), (1, 5))
    assert yt.shape == (B, 40, H, 36)
    np.testing.assert_allclose(be.host(Y), yt.detach().numpy(), atol=2e-5)
    s = be.host(SUMS)
    np.testing.assert_allclose(s[:40], yt.detach().sum((0, 2, 3)).numpy(), atol=1e-3)
    np.testing.assert_allclose(s[40:], (yt.detach() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-5)
    dy = rnd(rng, B, 40, H, 36)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    DY, DW25 = be.dev(dy), be.dev(np.ones((40, 25), np.float32))
    WS = be.zeros(int(be.lib.eegclip_tsconv_bwd_w_workspace_floats(B, H)))
    ok(be.lib.eegclip_tsconv_bwd_w(be.ptr(X), 64 * 250, 250, be.ptr(DY), be.ptr(DW25), be.ptr(WS), B, H, 250, 40, be.stream))
    np.testing.assert_allclose(be.host(DW25) - 1.0, wt.grad.numpy(), atol=2e-4 * max(1.0, np.abs(wt.grad.numpy()).max()))
    DX = be.dev(np.full((B, 64, 250), 7.0, np.float32))
    ok(be.lib.eegclip_tsconv_bwd_x(be.ptr(DY), be.ptr(W25), be.ptr(DX), 64 * 250, 250, B, H, 250, 40, be.stream))
    dx = be.host(DX)
    np.testing.assert_allclose(dx[:, :H], xt.grad.numpy()[:, :H], atol=3e-5)
    assert (dx[:, H:] == 7.0).all()                   # rows beyond H are not touched


@pytest.mark.parametrize("N,w", [(32, 1.0), (200, 0.99)])
def test_infonce_pieces_match_closed_form(be, N, w):
    rng = np.random.default_rng(N)
    D = 64
    a = rnd(rng, N, D, scale=2.0)
    b = rnd(rng, N, D)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    s = np.float32(math.log(1 / 0.07))
    raw = (a.astype(np.float64) @ b.T.astype(np.float64))
    X, SC = be.dev(raw.astype(np.float32)), be.dev(np.array([s], np.float32))
    LR, LC = be.zeros(N), be.zeros(N)
    ok(be.lib.eegclip_lse_rows(be.ptr(X), N, N, N, be.ptr(SC), be.ptr(LR), be.stream))
    ok(be.lib.eegclip_lse_cols(be.ptr(X), N, N, N, be.ptr(SC), be.ptr(LC), be.stream))
    S = torch.tensor(raw.astype(np.float32), dtype=torch.float64) * float(s)
    np.testing.assert_allclose(be.host(LR), torch.logsumexp(S, 1).numpy(), atol=2e-5)
    np.testing.assert_allclose(be.host(LC), torch.logsumexp(S, 0).numpy(), atol=2e-5)
    LOSS, DS, LOSS2 = be.zeros(1), be.zeros(1), be.zeros(1)
    ok(be.lib.eegclip_infonce_loss(be.ptr(X), N, N, be.ptr(SC), be.ptr(LR), be.ptr(LC), w, be.ptr(LOSS2), be.stream))
    ok(be.lib.eegclip_infonce_grad(be.ptr(X), N, N, N, 0, N, be.ptr(SC), be.ptr(LR), be.ptr(LC), w, be.ptr(LOSS), be.ptr(DS), be.stream))
    idx = torch.arange(N)
    lref = w * 0.5 * ((torch.logsumexp(S, 1) - S[idx, idx]).mean() + (torch.logsumexp(S, 0) - S[idx, idx]).mean())
    G = w * (torch.softmax(S, 1) + torch.softmax(S, 0) - 2 * torch.eye(N, dtype=torch.float64)) / (2 * N)
    assert abs(be.host(LOSS)[0] - float(lref)) < 2e-5 and abs(be.host(LOSS2)[0] - float(lref)) < 2e-5
    np.testing.assert_allclose(be.host(X), (float(s) * G).numpy(), atol=2e-6)
    assert abs(be.host(DS)[0] - float((G * torch.tensor(raw.astype(np.float32), dtype=torch.float64)).sum())) < 2e-5


def test_topk_and_count(be):
    rng = np.random.default_rng(11)
    rows, cols = 37, 200
    x = rnd(rng, rows, cols)
    x[3, 10] = x[3, 150] = 9.0                           # tie -> lowest index first
    X = be.dev(x)
    for k in (1, 5):
        OUT = be.zeros((rows, k), np.int64)
        ok(be.lib.eegclip_topk_rows(be.ptr(X), rows, cols, cols, k, None, be.ptr(OUT), be.stream))
        ref = np.argsort(-x, axis=1, kind="stable")[:, :k]
        assert (be.host(OUT) == ref).all()
    NEG, OUTN = be.dev(np.array([-2.0], np.float32)), be.zeros((rows, 1), np.int64)
    ok(be.lib.eegclip_topk_rows(be.ptr(X), rows, cols, cols, 1, be.ptr(NEG), be.ptr(OUTN), be.stream))
    assert (be.host(OUTN)[:, 0] == np.argmin(x, axis=1)).all()        # negative scale ranks by -x
    labels = ref[:, 0].copy()
    labels[::3] = (labels[::3] + 1) % cols
    CNT = be.zeros(1, np.int32)
    ok(be.lib.eegclip_count_equal(be.ptr(OUT), 5, be.ptr(be.dev(labels.astype(np.int64))), rows, be.ptr(CNT), be.stream))
    assert be.host(CNT)[0] == int((ref[:, 0] == labels).sum())


def test_gemm_silu_epilogue_and_prior_stage_kernels(be):
    import ctypes
    from eeg_image_decode_amd import _abi
    rng = np.random.default_rng(21)
    rows, cols, K = 37, 200, 24
    a, w, bias = rnd(rng, rows, K), rnd(rng, cols, K), rnd(rng, cols)
    A, W, BI, C, CP = be.dev(a), be.dev(w), be.dev(bias), be.zeros((rows, cols)), be.zeros((rows, cols))
    Dm = _abi.dim
    d = _abi.GemmDesc(M=rows, N=cols, K=K, A=be.ptr(A), Am=Dm(K), Ak=Dm(1), B=be.ptr(W), Bk=Dm(1), Bn=Dm(K), C=be.ptr(C), Cm=Dm(cols), Cn=Dm(1),
                      Cpre=be.ptr(CP), bias_n=be.ptr(BI), bias_m=None, R=None, Rm=Dm(0), Rn=Dm(0), alpha=1.0, accumulate=0, act=_abi.ACT_SILU,
                      drop_p=0.0, seed=0, drop_site=0, split_k=1)
    ok(be.lib.eegclip_gemm_f32(ctypes.byref(d), be.stream))
    pre = torch.tensor(a, dtype=torch.float64) @ torch.tensor(w, dtype=torch.float64).T + torch.tensor(bias, dtype=torch.float64)
    np.testing.assert_allclose(be.host(C), F.silu(pre).numpy(), atol=3e-5)
    # LayerNorm -> SiLU -> dropout and its backward pieces
    x, g, b, dy = rnd(rng, rows, cols), 1 + 0.1 * rnd(rng, cols), 0.1 * rnd(rng, cols), rnd(rng, rows, cols)
    X, G, Bt, DY = be.dev(x), be.dev(g), be.dev(b), be.dev(dy)
    YL, YA, MU, RS = be.zeros((rows, cols)), be.zeros((rows, cols)), be.zeros(rows), be.zeros(rows)
    p = 0.1
    ok(be.lib.eegclip_layernorm_silu_fwd(be.ptr(X), be.ptr(G), be.ptr(Bt), be.ptr(YL), be.ptr(YA), be.ptr(MU), be.ptr(RS), rows, cols, 1e-5, p, SEED, 4, be.stream))
    keep = torch.tensor(keep_mask(SEED, 4, rows * cols, p).reshape(rows, cols))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ln = F.layer_norm(xt, (cols,), torch.tensor(g, dtype=torch.float64), torch.tensor(b, dtype=torch.float64), 1e-5)
    ya = F.silu(ln) * keep / (1 - p)
    ya.backward(torch.tensor(dy, dtype=torch.float64))
    np.testing.assert_allclose(be.host(YL), ln.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(be.host(YA), ya.detach().numpy(), atol=3e-5)
    DLN, DX, DG, DB = be.zeros((rows, cols)), be.zeros((rows, cols)), be.zeros(cols), be.zeros(cols)
    ok(be.lib.eegclip_silu_bwd(be.ptr(DY), be.ptr(YL), be.ptr(DLN), rows * cols, 0, p, SEED, 4, be.stream))
    ok(be.lib.eegclip_layernorm_bwd(be.ptr(DLN), be.ptr(X), be.ptr(G), be.ptr(MU), be.ptr(RS), be.ptr(DX), be.ptr(DG), be.ptr(DB), rows, cols, 0, None, 0.0, 0, 0, be.stream))
    np.testing.assert_allclose(be.host(DX), xt.grad.numpy(), atol=5e-5)


def test_timestep_embedding_ddpm_and_mse(be):
    from oracle import prior as oprior
    rng = np.random.default_rng(22)
    n, dim = 9, 512
    t = np.array([0, 1, 5, 20, 333, 980, 999, 500, 7], np.float32)
    Tt, OUT = be.dev(t), be.zeros((n, dim))
    ok(be.lib.eegclip_timestep_embedding(be.ptr(Tt), n, dim, be.ptr(OUT), be.stream))
    np.testing.assert_allclose(be.host(OUT), oprior.timestep_embedding(torch.tensor(t), dim).numpy(), atol=3e-4)   # |t f| up to 999: fp32 sin/cos argument
    sch = oprior.DDPMSchedulerOracle()
    d = 1024
    h, nz = rnd(rng, n, d), rnd(rng, n, d)
    ts = t.astype(np.int64)
    SA, SB = be.dev(sch.alphas_cumprod.sqrt().numpy()), be.dev((1 - sch.alphas_cumprod).sqrt().numpy())
    H, NZ, TS, O2 = be.dev(h), be.dev(nz), be.dev(ts), be.zeros((n, d))
    ok(be.lib.eegclip_ddpm_add_noise(be.ptr(H), be.ptr(NZ), be.ptr(TS), be.ptr(SA), be.ptr(SB), be.ptr(O2), n, d, be.stream))
    np.testing.assert_allclose(be.host(O2), sch.add_noise(torch.tensor(h), torch.tensor(nz), torch.tensor(ts)).numpy(), atol=1e-6)
    # one ancestral step with CFG, against the oracle scheduler
    sch.set_timesteps(50)
    x, ec, eu, noise = rnd(rng, n, d), rnd(rng, n, d), rnd(rng, n, d), rnd(rng, n, d)
    for tstep in (980, 20, 0):
        sa, sb, c0, ct, sg = sch.step_coeffs(tstep)
        X, EC, EU, NO, O3, O4 = be.dev(x), be.dev(ec), be.dev(eu), be.dev(noise), be.zeros((n, d)), be.zeros((n, d))
        ok(be.lib.eegclip_ddpm_step(be.ptr(X), be.ptr(EC), be.ptr(EU), 5.0, sa, sb, c0, ct, sg, be.ptr(NO), be.ptr(O3), be.ptr(O4), n * d, be.stream))
        assert np.array_equal(be.host(O3), be.host(O4))               # out_dup: the second half of the next step's 2N-row input
        eps = torch.tensor(eu) + 5.0 * (torch.tensor(ec) - torch.tensor(eu))
        x0 = ((torch.tensor(x) - sb * eps) / sa).clamp(-1, 1)
        ref = c0 * x0 + ct * torch.tensor(x) + sg * torch.tensor(noise)
        np.testing.assert_allclose(be.host(O3), ref.numpy(), atol=2e-5)
    pred, tgt = rnd(rng, n, d), rnd(rng, n, d)
    PR, TG, LS, DP = be.dev(pred), be.dev(tgt), be.zeros(1), be.zeros((n, d))
    ok(be.lib.eegclip_mse_loss_grad(be.ptr(PR), be.ptr(TG), n * d, be.ptr(LS), be.ptr(DP), be.stream))
    assert abs(be.host(LS)[0] - float(((pred - tgt).astype(np.float64) ** 2).mean())) < 1e-5
    np.testing.assert_allclose(be.host(DP), 2 * (pred - tgt) / (n * d), atol=1e-9)


def _to16(a, f16):
    t = torch.tensor(a)
    t16 = t.to(torch.float16 if f16 else torch.bfloat16)
    return t16.view(torch.int16).numpy().copy(), t16.float().numpy()

@pytest.mark.parametrize("rows,cols,ce_rows", [(16, 1024, 8), (5, 64, 0), (3, 130, 3)])
def test_prior_stage_infer(be, rows, cols, ce_rows):
    """sampling-chain stage tail: SiLU(LayerNorm(x)) (+ skip) -> act_out; + time-embedding row + condition rows -> xin_out"""
    rng = np.random.default_rng(rows + cols)
    x, g, b_, sk, te, ce = rnd(rng, rows, cols), rnd(rng, cols), rnd(rng, cols), rnd(rng, rows, cols), rnd(rng, cols), rnd(rng, max(ce_rows, 1), cols)
    X, G, Bt, SK, TE, CE = be.dev(x), be.dev(g), be.dev(b_), be.dev(sk), be.dev(te), be.dev(ce)
    xt = torch.tensor(x, dtype=torch.float64)
    y = F.silu(F.layer_norm(xt, (cols,), torch.tensor(g, dtype=torch.float64), torch.tensor(b_, dtype=torch.float64), 1e-5)).numpy()
    for use_skip in (False, True):
        ACT, XIN = be.zeros((rows, cols)), be.zeros((rows, cols))
        ok(be.lib.eegclip_prior_stage_infer(be.ptr(X), be.ptr(G), be.ptr(Bt), be.ptr(SK) if use_skip else None, be.ptr(ACT), be.ptr(TE),
                                            be.ptr(CE) if ce_rows else None, ce_rows, be.ptr(XIN), rows, cols, 1e-5, be.stream))
        a = y + (sk if use_skip else 0)
        w = a + te
        w[:ce_rows] += ce[:ce_rows]
        np.testing.assert_allclose(be.host(ACT), a, atol=2e-5)
        np.testing.assert_allclose(be.host(XIN), w, atol=2e-5)
    ACT = be.zeros((rows, cols))
    ok(be.lib.eegclip_prior_stage_infer(be.ptr(X), be.ptr(G), be.ptr(Bt), None, be.ptr(ACT), None, None, 0, None, rows, cols, 1e-5, be.stream))
    np.testing.assert_allclose(be.host(ACT), y, atol=2e-5)
    assert be.lib.eegclip_prior_stage_infer(be.ptr(X), be.ptr(G), be.ptr(Bt), None, None, None, None, 0, None, rows, cols, 1e-5, be.stream) < 0


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("B,HW,heads,S,S_ip", [(2, 300, 3, 77, 4), (1, 64, 2, 77, 0), (1, 40, 1, 5, 16)])
def test_sdxl_cross_attention_with_ip_adapter_branch(be, f16, B, HW, heads, S, S_ip):
    from oracle import sdxl_attn
    rng = np.random.default_rng(B * 1000 + HW + S + S_ip + int(f16))
    C = heads * 64
    q16, qf = _to16(rnd(rng, B, HW, C), f16)
    k16, kf = _to16(rnd(rng, B, S, C), f16)
    v16, vf = _to16(rnd(rng, B, S, C), f16)
    Q, K, V, OUT = be.dev(q16), be.dev(k16), be.dev(v16), be.zeros((B, HW, C), np.int16)
    KI = VI = None
    kif = vif = None
    if S_ip:
        ki16, kif = _to16(rnd(rng, B, S_ip, C), f16)
        vi16, vif = _to16(rnd(rng, B, S_ip, C), f16)
        KI, VI = be.dev(ki16), be.dev(vi16)
    ok(be.lib.eegclip_cross_attn_fwd(be.ptr(Q), be.ptr(K), be.ptr(V), be.ptr(KI), be.ptr(VI), be.ptr(OUT), B, HW, heads, 64, S, S_ip, 0.75, int(f16), be.stream))
    got = torch.tensor(be.host(OUT)).view(torch.float16 if f16 else torch.bfloat16).float().numpy()
    ref = sdxl_attn.cross_attention(qf, kf, vf, heads, kif, vif, 0.75)
    tol = 6e-3 if f16 else 2.5e-2            # 16-bit probabilities and outputs: ~2^-11 (f16) / 2^-8 (bf16) relative
    np.testing.assert_allclose(got, ref, atol=tol)
    assert np.abs(got - ref).mean() < tol / 6


@pytest.mark.parametrize("B,H", [(3, 63), (2, 5), (33, 63)])
def test_fused_spatial_stage(be, B, H):
    """csrc/sconv.hip: BN1 -> ELU -> (H x 1) conv forward, weight gradient, and the two-pass input gradient + BatchNorm backward."""
    rng = np.random.default_rng(B * 100 + H)
    C, Wd = 40, 36
    y1 = rnd(rng, B, C, H, Wd) * 1.3 + 0.2
    g1, b1 = 1 + 0.1 * rnd(rng, C), 0.1 * rnd(rng, C)
    Ws, bs = (rnd(rng, C, C, H) / np.sqrt(C * H)).astype(np.float32), 0.1 * rnd(rng, C)
    dy2 = rnd(rng, B, C, Wd)
    yt = torch.tensor(y1, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(g1, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b1, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(Ws, dtype=torch.float64, requires_grad=True)
    z = F.elu(F.batch_norm(yt, None, None, gt, bt, True, 0.1, 1e-5))
    y2t = F.conv2d(z, wt.view(C, C, H, 1), torch.tensor(bs, dtype=torch.float64)).squeeze(2)
    y2t.backward(torch.tensor(dy2, dtype=torch.float64))
    mean = y1.astype(np.float64).mean((0, 2, 3))
    var = y1.astype(np.float64).var((0, 2, 3))
    Y1, G1, B1, WS, BS, DY2 = be.dev(y1), be.dev(g1), be.dev(b1), be.dev(Ws), be.dev(bs), be.dev(dy2)
    MU, RS = be.dev(mean.astype(np.float32)), be.dev((1 / np.sqrt(var + 1e-5)).astype(np.float32))
    # forward (exact fp32 products): K-slice partial tiles added into y2 with atomics | written to workspace slabs and summed by the statistics kernel
    for slabs in (False, True):
        Y2, S2 = (be.dev(np.full((B, C, Wd), np.nan, np.float32)) if slabs else be.zeros((B, C, Wd))), be.zeros(80, np.float64)
        WSF = be.dev(np.full(int(be.lib.eegclip_sconv_fwd_workspace_floats(B)), np.nan, np.float32)) if slabs else None
        ok(be.lib.eegclip_sconv_fwd(be.ptr(Y1), be.ptr(MU), be.ptr(RS), be.ptr(G1), be.ptr(B1), be.ptr(WS), be.ptr(BS), be.ptr(Y2), be.ptr(S2),
                                    B, H, 0, be.ptr(WSF) if slabs else None, be.stream))
        np.testing.assert_allclose(be.host(Y2), y2t.detach().numpy(), atol=5e-5)
        np.testing.assert_allclose(be.host(S2)[:40], y2t.detach().sum((0, 2)).numpy(), atol=1e-3)   # (sums of ~1e3 terms)
        np.testing.assert_allclose(be.host(S2)[40:], (y2t.detach() ** 2).sum((0, 2)).numpy(), rtol=1e-4)
    for precision in (_abi.PREC_F32, _abi.PREC_BF16X3):
        DWS = be.dev(np.ones((C, C, H), np.float32))
        WSP = be.zeros(int(be.lib.eegclip_sconv_bwd_w_workspace_floats(B, H)))
        ok(be.lib.eegclip_sconv_bwd_w(be.ptr(Y1), be.ptr(MU), be.ptr(RS), be.ptr(G1), be.ptr(B1), be.ptr(DY2), be.ptr(DWS), be.ptr(WSP), B, H, precision,
                                      be.stream))
        np.testing.assert_allclose(be.host(DWS) - 1.0, wt.grad.numpy(), atol=1e-4 * max(1.0, np.abs(wt.grad.numpy()).max()))
    # input gradient: exact fp32 products (no planes), then split-bf16 products with Ws^T pre-split by eegclip_split_rows
    K = C * H
    WH, WL = be.dev(np.full((K, 64), 0x7FC0, np.uint16)), be.dev(np.full((K, 64), 0x7FC0, np.uint16))
    it = (_abi.SplitItem * 1)(_abi.SplitItem(src=be.ptr(WS), hi=be.ptr(WH), lo=be.ptr(WL), rows=C, cols=K, ld_src=K, ld_out=64, transpose=1))
    ok(be.lib.eegclip_split_rows(it, 1, be.stream))
    for planes in ((None, None), (be.ptr(WH), be.ptr(WL))):
        SUMS, DY1, DG, DB = be.zeros(80, np.float64), be.zeros((B, C, H, Wd)), be.zeros(C), be.zeros(C)
        nws = int(be.lib.eegclip_sconv_bwd_x_stats_workspace_floats(B))
        WSX = be.dev(np.full(nws // 2, np.nan, np.float64)) if planes[0] else None       # (partial rows + column sums | atomics)
        ok(be.lib.eegclip_sconv_bwd_x_stats(be.ptr(DY2), be.ptr(WS), *planes, be.ptr(Y1), be.ptr(MU), be.ptr(RS), be.ptr(G1), be.ptr(B1), be.ptr(SUMS),
                                            be.ptr(WSX) if planes[0] else None, B, H, be.stream))
        ok(be.lib.eegclip_sconv_bwd_x_apply(be.ptr(DY2), be.ptr(WS), *planes, be.ptr(Y1), be.ptr(MU), be.ptr(RS), be.ptr(G1), be.ptr(B1), be.ptr(SUMS),
                                            None, float(B * H * Wd), be.ptr(DY1), be.ptr(DG), be.ptr(DB), B, H, be.stream))
        np.testing.assert_allclose(be.host(DY1), yt.grad.numpy(), atol=1e-6 + 2e-4 * np.abs(yt.grad.numpy()).max())
        np.testing.assert_allclose(be.host(DG), gt.grad.numpy(), atol=2e-4 * max(1.0, np.abs(gt.grad.numpy()).max()))
        np.testing.assert_allclose(be.host(DB), bt.grad.numpy(), atol=2e-4 * max(1.0, np.abs(bt.grad.numpy()).max()))
    assert be.lib.eegclip_sconv_bwd_x_stats(be.ptr(DY2), be.ptr(WS), be.ptr(WH), None, be.ptr(Y1), be.ptr(MU), be.ptr(RS), be.ptr(G1), be.ptr(B1),
                                            be.ptr(SUMS), None, B, H, be.stream) < 0


def _bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 128, 192)])
def test_logits_bf16(be, M, N, K):
    """csrc/logits_bf16.hip: features rounded to bf16 once, fp32 accumulate on v_mfma_f32_16x16x32_bf16, fp32 logits * scale"""
    rng = np.random.default_rng(M + N + K)
    a, b = rnd(rng, M, K), rnd(rng, N, K)
    A, B, SC = be.dev(a), be.dev(b), be.dev(np.array([2.6593], np.float32))
    A16, B16 = be.zeros((M, K // 2)), be.zeros((N, K // 2))          # bf16 storage viewed as float32 pairs
    ok(be.lib.eegclip_cast_bf16(be.ptr(A), be.ptr(A16), M * K, be.stream))
    ok(be.lib.eegclip_cast_bf16(be.ptr(B), be.ptr(B16), N * K, be.stream))
    got16 = be.host(A16).view(np.uint16).reshape(M, K)
    np.testing.assert_array_equal(got16, (_bf16_round(a).view(np.uint32) >> 16).astype(np.uint16))
    C = be.dev(np.full((M, N), np.nan, np.float32))
    ok(be.lib.eegclip_logits_bf16(be.ptr(A16), be.ptr(B16), be.ptr(C), M, N, K, N, be.ptr(SC), be.stream))
    ref = 2.6593 * _bf16_round(a).astype(np.float64) @ _bf16_round(b).astype(np.float64).T
    np.testing.assert_allclose(be.host(C), ref, atol=2e-4 * max(1.0, np.abs(ref).max()))
    assert be.lib.eegclip_logits_bf16(be.ptr(A16), be.ptr(B16), be.ptr(C), M, N - 1, K, N, be.ptr(SC), be.stream) < 0      # whole tiles only


@pytest.mark.parametrize("B,p", [(3, 0.0), (5, 0.5), (130, 0.25)])
def test_proj1x1_fused_stage(be, B, p):
    """csrc/proj1x1.hip: BN2 -> ELU -> dropout -> 1x1 conv -> (b, w*40+e) flatten, and the backward up to the BatchNorm statistics +
    bn_elu_bwd_apply, against autograd of the same torch ops with the Philox mask"""
    rng = np.random.default_rng(B)
    C, Wd = 40, 36
    y2 = rnd(rng, B, C, Wd) * 1.2 + 0.1
    g, bt, Wc, bc = 1 + 0.1 * rnd(rng, C), 0.1 * rnd(rng, C), rnd(rng, C, C, scale=0.2), 0.1 * rnd(rng, C)
    dfeat = rnd(rng, B, Wd * C)
    mean, var = y2.astype(np.float64).mean((0, 2)), y2.astype(np.float64).var((0, 2))
    keep = keep_mask(SEED, 2, B * C * Wd, p).reshape(B, C, Wd) if p > 0 else np.ones((B, C, Wd), bool)
    yt = torch.tensor(y2, dtype=torch.float64, requires_grad=True)
    gt, btt = torch.tensor(g, dtype=torch.float64, requires_grad=True), torch.tensor(bt, dtype=torch.float64, requires_grad=True)
    wt, bct = torch.tensor(Wc, dtype=torch.float64, requires_grad=True), torch.tensor(bc, dtype=torch.float64, requires_grad=True)
    z = F.elu(F.batch_norm(yt, None, None, gt, btt, True, 0.1, 1e-5)) * torch.tensor(keep) / (1 - p)
    ft = (torch.einsum("ec,bcw->bwe", wt, z) + bct).reshape(B, Wd * C)
    ft.backward(torch.tensor(dfeat, dtype=torch.float64))
    Y2, MU, RS = be.dev(y2), be.dev(mean.astype(np.float32)), be.dev((1 / np.sqrt(var + 1e-5)).astype(np.float32))
    G, BT, WC, BC, DF = be.dev(g), be.dev(bt), be.dev(Wc), be.dev(bc), be.dev(dfeat)
    Z2, FEAT = be.zeros((B, C, Wd)), be.zeros((B, Wd * C))
    ok(be.lib.eegclip_proj1x1_fwd(be.ptr(Y2), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(BT), be.ptr(WC), be.ptr(BC), be.ptr(Z2), be.ptr(FEAT), B, p,
                                  SEED, 2, be.stream))
    np.testing.assert_allclose(be.host(Z2), z.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(be.host(FEAT), ft.detach().numpy(), atol=5e-5)
    # the same with the BatchNorm finalize folded into the prologue: statistics from per-sample partial rows [sum | sumsq], mean / rstd / running statistics
    # written by workgroup 0 (eegclip_proj1x1_fwd_rows)
    rows = np.concatenate([y2.astype(np.float64).sum(2), (y2.astype(np.float64) ** 2).sum(2)], axis=1)
    ROWS = be.dev(rows)
    MU2, RS2 = be.dev(np.full(C, np.nan, np.float32)), be.dev(np.full(C, np.nan, np.float32))
    RM, RV, NBT = be.dev(np.full(C, 0.5, np.float32)), be.dev(np.full(C, 2.0, np.float32)), be.dev(np.array([3], np.int64))
    Z2r, FEATr = be.zeros((B, C, Wd)), be.zeros((B, Wd * C))
    count = float(B * Wd)
    ok(be.lib.eegclip_proj1x1_fwd_rows(be.ptr(Y2), be.ptr(ROWS), B, count, 1e-5, 0.1, be.ptr(MU2), be.ptr(RS2), be.ptr(RM), be.ptr(RV), be.ptr(NBT), be.ptr(G),
                                       be.ptr(BT), be.ptr(WC), be.ptr(BC), be.ptr(Z2r), be.ptr(FEATr), B, p, SEED, 2, be.stream))
    np.testing.assert_allclose(be.host(Z2r), z.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(be.host(FEATr), ft.detach().numpy(), atol=5e-5)
    np.testing.assert_allclose(be.host(MU2), mean, atol=1e-6)
    np.testing.assert_allclose(be.host(RS2), 1 / np.sqrt(var + 1e-5), rtol=1e-5)
    np.testing.assert_allclose(be.host(RM), 0.45 + 0.1 * mean, atol=1e-6)
    np.testing.assert_allclose(be.host(RV), 1.8 + 0.1 * var * (count / (count - 1) if count > 1 else 1.0), rtol=1e-5)
    assert int(be.host(NBT)[0]) == 4
    assert be.lib.eegclip_proj1x1_fwd_rows(be.ptr(Y2), None, B, count, 1e-5, 0.1, be.ptr(MU2), be.ptr(RS2), None, None, None, be.ptr(G), be.ptr(BT), be.ptr(WC),
                                           be.ptr(BC), be.ptr(Z2r), be.ptr(FEATr), B, p, SEED, 2, be.stream) < 0
    sums_seen = []
    for use_ws in (False, True):        # atomics on dW / dbias / sums, then per-workgroup partial rows + column reduction
        DZ2, DW, DBC, SUMS = be.zeros((B, C, Wd)), be.dev(np.ones((C, C), np.float32)), be.zeros(C), be.zeros(2 * C, np.float64)
        nws = int(be.lib.eegclip_proj1x1_bwd_workspace_floats(B))
        assert nws == 2 * B * (C * C + 3 * C)
        WSP = be.dev(np.full(nws // 2, np.nan, np.float64)) if use_ws else None
        ok(be.lib.eegclip_proj1x1_bwd(be.ptr(DF), be.ptr(Z2), be.ptr(WC), be.ptr(Y2), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(BT), be.ptr(DZ2), be.ptr(DW),
                                      be.ptr(DBC), be.ptr(SUMS), be.ptr(WSP) if use_ws else None, B, p, SEED, 2, be.stream))
        np.testing.assert_allclose(be.host(DW) - 1.0, wt.grad.numpy(), atol=2e-4 * max(1.0, np.abs(wt.grad.numpy()).max()))
        np.testing.assert_allclose(be.host(DBC), bct.grad.numpy(), atol=2e-4 * max(1.0, np.abs(bct.grad.numpy()).max()))
        sums_seen.append(be.host(SUMS).copy())
    np.testing.assert_allclose(sums_seen[1], sums_seen[0], rtol=1e-9, atol=1e-9)
    DY2, DG, DB = be.zeros((B, C, Wd)), be.zeros(C), be.zeros(C)
    ok(be.lib.eegclip_bn_elu_bwd_apply(be.ptr(DZ2), be.ptr(Y2), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(BT), be.ptr(SUMS), None, float(B * Wd),
                                       be.ptr(DY2), be.ptr(DG), be.ptr(DB), B, C, Wd, p, SEED, 2, be.stream))
    np.testing.assert_allclose(be.host(DY2), yt.grad.numpy(), atol=1e-6 + 3e-4 * np.abs(yt.grad.numpy()).max())
    np.testing.assert_allclose(be.host(DG), gt.grad.numpy(), atol=3e-4 * max(1.0, np.abs(gt.grad.numpy()).max()))
    np.testing.assert_allclose(be.host(DB), btt.grad.numpy(), atol=3e-4 * max(1.0, np.abs(btt.grad.numpy()).max()))
    # ---- round 6: feat also as bf16 planes + dense plane splits RIDING in the launch (extra workgroups); the backward from K-parallel GEMM slabs, its
    # BatchNorm sums as a compact per-sample table that the apply pass adds itself, the dW / dbias reduction as a launch of its own
    from test_kernels_wgrad import split
    FH, FL = be.zeros((B, Wd * C), np.uint16), be.zeros((B, Wd * C), np.uint16)
    ra, rb = rnd(rng, 3, 64), rnd(rng, 1, 1000)
    RA, RB = be.dev(ra), be.dev(rb)
    RAH, RAL, RBH, RBL = be.zeros((3, 64), np.uint16), be.zeros((3, 64), np.uint16), be.zeros((1, 1000), np.uint16), be.zeros((1, 1000), np.uint16)
    riders = (_abi.SplitItem * 2)(_abi.SplitItem(src=be.ptr(RA), hi=be.ptr(RAH), lo=be.ptr(RAL), rows=3, cols=64, ld_src=64, ld_out=64, transpose=0),
                                  _abi.SplitItem(src=be.ptr(RB), hi=be.ptr(RBH), lo=be.ptr(RBL), rows=1, cols=1000, ld_src=1000, ld_out=1000, transpose=0))
    Z2p, FEATp = be.zeros((B, C, Wd)), be.zeros((B, Wd * C))
    ok(be.lib.eegclip_proj1x1_fwd_rows_planes(be.ptr(Y2), None, 0, 1.0, 1e-5, 0.1, be.ptr(MU), be.ptr(RS), None, None, None, be.ptr(G), be.ptr(BT), be.ptr(WC),
                                              be.ptr(BC), be.ptr(Z2p), be.ptr(FEATp), B, p, SEED, 2, be.ptr(FH), be.ptr(FL), riders, 2, be.stream))
    np.testing.assert_array_equal(be.host(FEATp), be.host(FEAT))
    np.testing.assert_array_equal(be.host(Z2p), be.host(Z2))
    u16 = lambda v: (np.asarray(v).view(np.uint32) >> 16).astype(np.uint16)
    for src, H, Lo in ((be.host(FEAT), FH, FL), (ra, RAH, RAL), (rb, RBH, RBL)):
        h2, l2 = split(src)
        np.testing.assert_array_equal(be.host(H), u16(h2))
        np.testing.assert_array_equal(be.host(Lo), u16(l2))
    bad = (_abi.SplitItem * 1)(_abi.SplitItem(src=be.ptr(RA), hi=be.ptr(RAH), lo=be.ptr(RAL), rows=3, cols=64, ld_src=64, ld_out=64, transpose=1))
    assert be.lib.eegclip_proj1x1_fwd_rows_planes(be.ptr(Y2), None, 0, 1.0, 1e-5, 0.1, be.ptr(MU), be.ptr(RS), None, None, None, be.ptr(G), be.ptr(BT), be.ptr(WC),
                                                  be.ptr(BC), be.ptr(Z2p), be.ptr(FEATp), B, p, SEED, 2, be.ptr(FH), be.ptr(FL), bad, 1, be.stream) < 0
    n_sl, stride = 3, B * Wd * C + 8
    parts = rnd(rng, n_sl, B, Wd * C)
    parts[n_sl - 1] = dfeat - parts[:n_sl - 1].sum(0)                    # the slabs add up to (nearly) the same upstream gradient
    tot = parts[0].copy()
    for i in range(1, n_sl):
        tot = tot + parts[i]
    slab_buf = np.full(n_sl * stride, np.nan, np.float32)
    for i in range(n_sl):
        slab_buf[i * stride:i * stride + B * Wd * C] = parts[i].ravel()
    SL, TOT = be.dev(slab_buf), be.dev(tot)
    DZa, DWa, DBa, SUMa = be.zeros((B, C, Wd)), be.zeros((C, C)), be.zeros(C), be.zeros(2 * C, np.float64)
    WSa = be.dev(np.full(nws // 2, np.nan, np.float64))
    ok(be.lib.eegclip_proj1x1_bwd(be.ptr(TOT), be.ptr(Z2), be.ptr(WC), be.ptr(Y2), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(BT), be.ptr(DZa), be.ptr(DWa),
                                  be.ptr(DBa), be.ptr(SUMa), be.ptr(WSa), B, p, SEED, 2, be.stream))
    DYa, DGa, DBta = be.zeros((B, C, Wd)), be.zeros(C), be.zeros(C)
    ok(be.lib.eegclip_bn_elu_bwd_apply(be.ptr(DZa), be.ptr(Y2), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(BT), be.ptr(SUMa), None, float(B * Wd),
                                       be.ptr(DYa), be.ptr(DGa), be.ptr(DBta), B, C, Wd, p, SEED, 2, be.stream))
    DZb, DWb, DBb = be.zeros((B, C, Wd)), be.zeros((C, C)), be.zeros(C)
    WSb, BNR = be.dev(np.full(nws // 2, np.nan, np.float64)), be.dev(np.full((B, 2 * C), np.nan, np.float64))
    ok(be.lib.eegclip_proj1x1_bwd_rows(be.ptr(SL), n_sl, stride, be.ptr(Z2), be.ptr(WC), be.ptr(Y2), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(BT), be.ptr(DZb),
                                       be.ptr(WSb), be.ptr(BNR), B, p, SEED, 2, be.stream))
    DYb, DGb, DBtb = be.zeros((B, C, Wd)), be.dev(np.full(C, 0.25, np.float32)), be.dev(np.full(C, -0.25, np.float32))
    ok(be.lib.eegclip_bn_elu_bwd_apply_rows(be.ptr(DZb), be.ptr(Y2), be.ptr(MU), be.ptr(RS), be.ptr(G), be.ptr(BT), be.ptr(BNR), B, 2 * C, 0, float(B * Wd),
                                            be.ptr(DYb), be.ptr(DGb), be.ptr(DBtb), B, C, Wd, p, SEED, 2, be.stream))
    ok(be.lib.eegclip_proj1x1_bwd_reduce(be.ptr(WSb), B, be.ptr(DWb), be.ptr(DBb), None, be.stream))
    np.testing.assert_array_equal(be.host(DZb), be.host(DZa))
    np.testing.assert_allclose(be.host(BNR).sum(0), be.host(SUMa), rtol=1e-9, atol=1e-9)
    scale = max(1.0, float(np.abs(be.host(DYa)).max()))
    np.testing.assert_allclose(be.host(DYb), be.host(DYa), atol=2e-6 * scale)
    np.testing.assert_allclose(be.host(DGb) - 0.25, be.host(DGa), atol=1e-5 * max(1.0, float(np.abs(be.host(DGa)).max())))
    np.testing.assert_allclose(be.host(DBtb) + 0.25, be.host(DBta), atol=1e-5 * max(1.0, float(np.abs(be.host(DBta)).max())))
    np.testing.assert_allclose(be.host(DWb), be.host(DWa), atol=1e-5 * max(1.0, float(np.abs(be.host(DWa)).max())))
    np.testing.assert_allclose(be.host(DBb), be.host(DBa), atol=1e-5 * max(1.0, float(np.abs(be.host(DBa)).max())))


@pytest.mark.parametrize("rows,cols,neg", [(256, 1654, False), (5, 7, True), (64, 200, False)])
def test_top1_count(be, rows, cols, neg):
    """eegclip_top1_count == arg-max per row (ties -> lowest index, ranking by scale * x) compared with the labels, counted"""
    rng = np.random.default_rng(rows + cols)
    x = rnd(rng, rows, cols)
    x[0, :] = 1.0                                          # a row of ties: index 0 (or the lowest index under a negative scale, too)
    x[1 % rows, [2 % cols, 5 % cols]] = 9.0
    labels = rng.integers(0, cols, size=rows).astype(np.int64)
    sc = np.array([-2.0 if neg else 2.5], np.float32)
    ref = np.argmax(sc[0] * x, axis=1)
    labels[::3] = ref[::3]
    X, LAB, SC, CNT = be.dev(x), be.dev(labels), be.dev(sc), be.dev(np.array([7], np.int32))
    ok(be.lib.eegclip_top1_count(be.ptr(X), rows, cols, cols, be.ptr(SC), be.ptr(LAB), be.ptr(CNT), be.stream))
    assert int(be.host(CNT)[0]) == 7 + int((ref == labels).sum())
    assert be.lib.eegclip_top1_count(be.ptr(X), rows, cols, cols - 1, be.ptr(SC), be.ptr(LAB), be.ptr(CNT), be.stream) < 0
