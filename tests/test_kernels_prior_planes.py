"""The plane-emitting pieces of the diffusion prior's training plans on both backends: the fused stage-tail kernels against the launches they replace
(eegclip_layernorm_silu_fwd + skip add; eegclip_silu_bwd + eegclip_layernorm_bwd), eegclip_silu_bwd_planes, eegclip_split_transpose."""
import numpy as np
import pytest

from backends import be  # noqa: F401
from eeg_image_decode_amd import _abi
from test_kernels_wgrad import bf16_round


def f32_of(u16):
    return (np.asarray(u16).astype(np.uint32) << 16).view(np.float32)


def check_planes(hi, lo, value):
    h = f32_of(hi)
    np.testing.assert_array_equal(h, bf16_round(value))
    np.testing.assert_array_equal(f32_of(lo), bf16_round(value - h))


@pytest.mark.parametrize("rows,cols,p,skip", [(8, 64, 0.0, 0), (10, 128, 0.25, 1), (6, 1024, 0.1, 1), (5, 320, 0.0, 0)])
def test_stage_forward_with_skip_and_planes(be, rows, cols, p, skip):
    rng = np.random.default_rng(rows + cols)
    x = rng.standard_normal((rows, cols)).astype(np.float32) * 2 + 0.3
    g, b = rng.standard_normal(cols).astype(np.float32), rng.standard_normal(cols).astype(np.float32)
    sk = rng.standard_normal((rows, cols)).astype(np.float32)
    X, G, B, SK = be.dev(x), be.dev(g), be.dev(b), be.dev(sk)
    ln0, act0, mu0, rs0 = be.zeros((rows, cols)), be.zeros((rows, cols)), be.zeros(rows), be.zeros(rows)
    assert be.lib.eegclip_layernorm_silu_fwd(be.ptr(X), be.ptr(G), be.ptr(B), be.ptr(ln0), be.ptr(act0), be.ptr(mu0), be.ptr(rs0), rows, cols, 1e-5, p, 77, 3, be.stream) == 0
    ln1, act1, mu1, rs1 = be.zeros((rows, cols)), be.zeros((rows, cols)), be.zeros(rows), be.zeros(rows)
    hi, lo = be.dev(np.zeros((rows, cols), np.uint16)), be.dev(np.zeros((rows, cols), np.uint16))
    assert be.lib.eegclip_prior_stage_fwd(be.ptr(X), be.ptr(G), be.ptr(B), be.ptr(SK) if skip else None, be.ptr(ln1), be.ptr(act1), be.ptr(mu1), be.ptr(rs1),
                                          be.ptr(hi), be.ptr(lo), rows, cols, 1e-5, p, 77, 3, be.stream) == 0
    be.sync()
    # (the fused kernel gives a lane 4 consecutive columns: another summation order for the statistics than the per-element kernel -- rounding-level
    #  differences; a wrong dropout mask would be an O(1) difference)
    np.testing.assert_allclose(be.host(ln1), be.host(ln0), atol=3e-6 * float(np.abs(be.host(ln0)).max()))
    np.testing.assert_allclose(be.host(mu1), be.host(mu0), atol=1e-6)
    np.testing.assert_allclose(be.host(rs1), be.host(rs0), rtol=1e-5)
    want = be.host(act0) + (sk if skip else 0.0)
    np.testing.assert_allclose(be.host(act1), want, atol=4e-6 * max(1.0, float(np.abs(want).max())))
    check_planes(be.host(hi), be.host(lo), be.host(act1))


@pytest.mark.parametrize("rows,cols,p,planes", [(16, 64, 0.0, 1), (24, 128, 0.25, 1), (9, 1024, 0.1, 1), (13, 320, 0.0, 0)])
def test_stage_backward_equals_silu_bwd_plus_layernorm_bwd(be, rows, cols, p, planes):
    rng = np.random.default_rng(rows + cols + 1)
    x = rng.standard_normal((rows, cols)).astype(np.float32) * 1.5 - 0.2
    g, b = (1 + 0.3 * rng.standard_normal(cols)).astype(np.float32), rng.standard_normal(cols).astype(np.float32)
    dact = rng.standard_normal((rows, cols)).astype(np.float32)
    X, G, B, DA = be.dev(x), be.dev(g), be.dev(b), be.dev(dact)
    ln, act, mu, rs = be.zeros((rows, cols)), be.zeros((rows, cols)), be.zeros(rows), be.zeros(rows)
    assert be.lib.eegclip_layernorm_silu_fwd(be.ptr(X), be.ptr(G), be.ptr(B), be.ptr(ln), be.ptr(act), be.ptr(mu), be.ptr(rs), rows, cols, 1e-5, p, 5, 2, be.stream) == 0
    # the two launches of the unfused plan
    dln, dx0, dg0, db0 = be.zeros((rows, cols)), be.zeros((rows, cols)), be.zeros(cols), be.zeros(cols)
    assert be.lib.eegclip_silu_bwd(be.ptr(DA), be.ptr(ln), be.ptr(dln), rows * cols, 0, p, 5, 2, be.stream) == 0
    assert be.lib.eegclip_layernorm_bwd(be.ptr(dln), be.ptr(X), be.ptr(G), be.ptr(mu), be.ptr(rs), be.ptr(dx0), be.ptr(dg0), be.ptr(db0), rows, cols, 0, None, 0.0,
                                        0, 0, be.stream) == 0
    dx1, dg1, db1 = be.zeros((rows, cols)), be.zeros(cols), be.zeros(cols)
    hi, lo = be.dev(np.zeros((rows, cols), np.uint16)), be.dev(np.zeros((rows, cols), np.uint16))
    assert be.lib.eegclip_prior_stage_bwd(be.ptr(DA), be.ptr(ln), be.ptr(X), be.ptr(G), be.ptr(mu), be.ptr(rs), be.ptr(dx1), be.ptr(hi) if planes else None,
                                          be.ptr(lo) if planes else None, be.ptr(dg1), be.ptr(db1), rows, cols, p, 5, 2, None, be.stream) == 0
    be.sync()
    sc = float(np.abs(be.host(dx0)).max())
    np.testing.assert_allclose(be.host(dx1), be.host(dx0), atol=3e-6 * max(1.0, sc))
    np.testing.assert_allclose(be.host(dg1), be.host(dg0), atol=1e-5 * max(1.0, float(np.abs(be.host(dg0)).max())))
    np.testing.assert_allclose(be.host(db1), be.host(db0), atol=1e-5 * max(1.0, float(np.abs(be.host(db0)).max())))
    if planes:
        check_planes(be.host(hi), be.host(lo), be.host(dx1))
    # planes only (no fp32 copy) is accepted too
    if planes:
        # ... and the parameter gradients through per-workgroup partial rows + their own summing launch (what the plans use)
        dg2, db2 = be.dev(np.ones(cols, np.float32)), be.dev(np.full(cols, 2.0, np.float32))
        ws = be.dev(np.full(int(be.lib.eegclip_prior_stage_bwd_workspace_floats(rows, cols)), np.nan, np.float32))
        assert be.lib.eegclip_prior_stage_bwd(be.ptr(DA), be.ptr(ln), be.ptr(X), be.ptr(G), be.ptr(mu), be.ptr(rs), None, be.ptr(hi), be.ptr(lo), None,
                                              None, rows, cols, p, 5, 2, be.ptr(ws), be.stream) == 0
        assert be.lib.eegclip_prior_stage_bwd_params(be.ptr(ws), rows, cols, be.ptr(dg2), be.ptr(db2), be.stream) == 0
        be.sync()
        check_planes(be.host(hi), be.host(lo), be.host(dx1))
        np.testing.assert_allclose(be.host(dg2) - 1.0, be.host(dg0), atol=1e-5 * max(1.0, float(np.abs(be.host(dg0)).max())))
        np.testing.assert_allclose(be.host(db2) - 2.0, be.host(db0), atol=1e-5 * max(1.0, float(np.abs(be.host(db0)).max())))


def test_silu_backward_as_planes(be):
    rng = np.random.default_rng(4)
    n = 4 * 333
    dy, pre = rng.standard_normal(n).astype(np.float32), (2 * rng.standard_normal(n)).astype(np.float32)
    DY, PRE, dx = be.dev(dy), be.dev(pre), be.zeros(n)
    hi, lo = be.dev(np.zeros(n, np.uint16)), be.dev(np.zeros(n, np.uint16))
    assert be.lib.eegclip_silu_bwd(be.ptr(DY), be.ptr(PRE), be.ptr(dx), n, 0, 0.0, 0, 0, be.stream) == 0
    assert be.lib.eegclip_silu_bwd_planes(be.ptr(DY), be.ptr(PRE), be.ptr(hi), be.ptr(lo), n, be.stream) == 0
    be.sync()
    check_planes(be.host(hi), be.host(lo), be.host(dx))
    assert be.lib.eegclip_silu_bwd_planes(be.ptr(DY), be.ptr(PRE), be.ptr(hi), be.ptr(lo), n + 2, be.stream) != 0


def test_split_transpose_items(be):
    rng = np.random.default_rng(6)
    shapes = [(64, 64, 64, 64), (128, 64, 80, 136), (64, 192, 192, 64)]          # rows, cols, ld_src, ld_out
    items = (_abi.SplitItem * len(shapes))()
    keep = []
    for i, (r, c, lds, ldo) in enumerate(shapes):
        w = rng.standard_normal((r, lds)).astype(np.float32)
        W, hi, lo = be.dev(w), be.dev(np.full((c, ldo), 0x7FC0, np.uint16)), be.dev(np.full((c, ldo), 0x7FC0, np.uint16))
        keep.append((w, W, hi, lo, r, c))
        items[i] = _abi.SplitItem(src=be.ptr(W), hi=be.ptr(hi), lo=be.ptr(lo), rows=r, cols=c, ld_src=lds, ld_out=ldo, transpose=1)
    assert be.lib.eegclip_split_transpose(items, len(shapes), be.stream) == 0
    be.sync()
    for w, W, hi, lo, r, c in keep:
        check_planes(be.host(hi)[:, :r], be.host(lo)[:, :r], w[:, :c].T.copy())
        assert (np.asarray(be.host(hi))[:, r:] == 0x7FC0).all()
    items[0].rows = 96
    assert be.lib.eegclip_split_transpose(items, 1, be.stream) != 0


def test_stage_kernels_and_embed_pack_reject_bad_arguments(be):
    x = be.zeros((8, 64))
    v = be.zeros(64)
    r = be.zeros(8)
    hi = be.dev(np.zeros((8, 64), np.uint16))
    L = be.lib
    # planes: both or neither
    assert L.eegclip_prior_stage_fwd(be.ptr(x), be.ptr(v), be.ptr(v), None, be.ptr(x), be.ptr(x), be.ptr(r), be.ptr(r), be.ptr(hi), None, 8, 64, 1e-5, 0.0, 0, 0, be.stream) != 0
    assert L.eegclip_prior_stage_fwd(be.ptr(x), be.ptr(v), be.ptr(v), None, be.ptr(x), be.ptr(x), be.ptr(r), be.ptr(r), None, None, 8, 64, 1e-5, 1.0, 0, 0, be.stream) != 0      # p = 1
    # backward: an output is required; parameter gradients need a destination (arrays or workspace)
    assert L.eegclip_prior_stage_bwd(be.ptr(x), be.ptr(x), be.ptr(x), be.ptr(v), be.ptr(r), be.ptr(r), None, None, None, be.ptr(v), be.ptr(v), 8, 64, 0.0, 0, 0, None, be.stream) != 0
    assert L.eegclip_prior_stage_bwd(be.ptr(x), be.ptr(x), be.ptr(x), be.ptr(v), be.ptr(r), be.ptr(r), be.ptr(x), None, None, None, None, 8, 64, 0.0, 0, 0, None, be.stream) != 0
    assert int(L.eegclip_prior_stage_bwd_workspace_floats(0, 64)) == 0 and int(L.eegclip_prior_stage_bwd_workspace_floats(16, 64)) == 2 * 2 * 64
    # the per-subject value-embedding pack: stride at least one matrix
    w = be.zeros(250 * 250)
    out = be.dev(np.zeros(int(L.eegclip_token_block_packed_embed_bytes(1)) // 2, np.uint16))
    assert L.eegclip_token_block_pack_embed(be.ptr(w), 250 * 250, 1, be.ptr(out), be.stream) == 0
    assert L.eegclip_token_block_pack_embed(be.ptr(w), 100, 1, be.ptr(out), be.stream) != 0
    assert L.eegclip_token_block_pack_embed(be.ptr(w), 250 * 250, 0, be.ptr(out), be.stream) != 0
    assert int(L.eegclip_token_block_packed_embed_bytes(0)) == 0
