"""csrc/wgrad_tok.hip through the C ABI on both backends: the token-plane layout (eegclip_tok_planes_from_f32) and the weight-gradient GEMM over it
(eegclip_wgrad_tok: LDS-DMA staged k-tiles, LDS transpose reads, split-bf16 products, slab reduction in slice order) against numpy on the SAME split
operands -- every shape / map the backward plan uses: plain, 62-per-head rows and columns, three channel groups, both bias routes, several
problems per launch."""
import ctypes

import numpy as np
import pytest

from backends import BACKENDS, be  # noqa: F401
from eeg_image_decode_amd import _abi


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_round(x)
    return hi, bf16_round(x - hi)


def planes_to_f32(blob, B):
    """token planes (uint16 view of B blocks [2][64][256]) -> (hi, lo) fp32 arrays (B * 64, 256)"""
    p = np.asarray(blob).view(np.uint16).reshape(B, 2, 64, 256).astype(np.uint32) << 16
    f = p.view(np.float32)
    return f[:, 0].reshape(B * 64, 256), f[:, 1].reshape(B * 64, 256)


def head_channels(cols):
    """column 62 head + d -> channel 64 head + d"""
    c = np.arange(cols)
    return 64 * (c // 62) + c % 62


def make_planes(be, x, heads, ones):
    rows, cols = x.shape
    B = rows // 64
    dst = be.dev(np.full(B * 32768, 0x7FC0, np.uint16))               # NaN-filled: every element must be written
    X = be.dev(x)
    assert be.lib.eegclip_tok_planes_from_f32(be.ptr(X), cols, rows, cols, int(heads), int(ones), be.ptr(dst), be.stream) == 0
    be.sync()
    return dst


@pytest.mark.parametrize("cols,heads,ones", [(250, 0, 1), (256, 0, 0), (248, 1, 1), (250, 0, 0)])
def test_token_planes_from_fp32(be, cols, heads, ones):
    rng = np.random.default_rng(5)
    B = 2
    x = rng.standard_normal((64 * B, cols)).astype(np.float32)
    hi, lo = planes_to_f32(be.host(make_planes(be, x, heads, ones)), B)
    ch = head_channels(cols) if heads else np.arange(cols)
    eh, el = np.zeros((64 * B, 256), np.float32), np.zeros((64 * B, 256), np.float32)
    eh[:, ch], el[:, ch] = split(x)
    if ones:
        eh[:, 255] = 1.0
    np.testing.assert_array_equal(hi, eh)
    np.testing.assert_array_equal(lo, el)


def reference(dy, x):
    """dW = dY^T X with the three split products, fp64 accumulation"""
    ah, al = (v.astype(np.float64) for v in split(dy))
    bh, bl = (v.astype(np.float64) for v in split(x))
    return ah.T @ bh + ah.T @ bl + al.T @ bh, (ah + al).sum(0)


CASES = {
    # name: (M, N, m_groups, heads_m, heads_n, bias route)        bias route: 0 none, 1 ones column, 2 all-ones fragment
    "ffn2": (250, 256, 1, 0, 0, 2),
    "ffn1": (256, 250, 1, 0, 0, 1),
    "out_proj": (250, 248, 1, 0, 1, 1),
    "qkv": (744, 250, 3, 1, 0, 1),
    "embed": (250, 250, 1, 0, 0, 1),
    "nobias": (250, 250, 1, 0, 0, 0),
}


def run_problems(be, names, B, slices, variant, seed=0):
    rng = np.random.default_rng(seed)
    probs = (_abi.WgradTokProblem * len(names))()
    keep, outs = [], []
    for i, nm in enumerate(names):
        M, N, mg, hm, hn, br = CASES[nm]
        dy = (rng.standard_normal((64 * B, M)) * rng.uniform(0.5, 2.0)).astype(np.float32)
        x = rng.standard_normal((64 * B, N)).astype(np.float32)
        per = M // mg
        a = [make_planes(be, dy[:, g * per:(g + 1) * per], hm, 0) for g in range(mg)]
        if mg > 1:                                            # the channel groups of one operand: consecutive regions, a_group_stride apart
            blob = be.dev(np.concatenate([be.host(v) for v in a]))
            a = [blob]
        b = make_planes(be, x, hn, br == 1)
        out0 = rng.standard_normal((M, N + 3)).astype(np.float32)
        bias0 = rng.standard_normal(M).astype(np.float32)
        out, bias = be.dev(out0), be.dev(bias0)
        keep += [a, b, out, bias]
        probs[i] = _abi.WgradTokProblem(a=be.ptr(a[0]), b=be.ptr(b), a_group_stride=B * 65536 if mg > 1 else 0, m_groups=mg, heads_m=hm, heads_n=hn, M=M, N=N,
                                        out=be.ptr(out), ldo=N + 3, bias_out=be.ptr(bias) if br else None, bias_mfma=int(br == 2))
        ew, eb = reference(dy, x)
        outs.append((nm, out, bias, out0, bias0, ew, eb, br))
    nws = int(be.lib.eegclip_wgrad_tok_workspace_floats(probs, len(names), B, slices))
    assert nws > 0
    ws = be.dev(np.full(nws, np.nan, np.float32))             # every slab element that is read must have been written
    assert be.lib.eegclip_wgrad_tok(probs, len(names), B, slices, be.ptr(ws), variant, be.stream) == 0
    assert be.lib.eegclip_wgrad_tok_reduce(probs, len(names), B, slices, be.ptr(ws), be.stream) == 0
    be.sync()
    for nm, out, bias, out0, bias0, ew, eb, br in outs:
        got = be.host(out).astype(np.float64)
        N = ew.shape[1]
        scale = np.abs(ew).max()
        np.testing.assert_allclose(got[:, :N] - out0[:, :N], ew, atol=2e-6 * scale * np.sqrt(64 * B), err_msg=nm)
        np.testing.assert_array_equal(got[:, N:], out0[:, N:])                    # the padding columns of `out` are untouched
        gb = be.host(bias).astype(np.float64) - bias0
        if br:
            np.testing.assert_allclose(gb, eb, atol=2e-6 * np.abs(eb).max() * np.sqrt(64 * B) + 1e-5, err_msg=nm + " bias")
        else:
            np.testing.assert_array_equal(gb, 0.0)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("name", list(CASES))
def test_wgrad_tok_single_problem(be, name, variant):
    run_problems(be, [name], 2, 1, variant)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_wgrad_tok_grouped_launch_and_slices(be, variant):
    run_problems(be, ["ffn2", "ffn1", "out_proj"], 4, 2, variant, seed=3)       # 8 k-tiles in 2 slices: 4 per workgroup = the ring's full depth


@pytest.mark.parametrize("slices,step", [(1, 1), (3, 7)])
def test_wgrad_tok_reduce_that_steps_the_optimizer(be, slices, step):
    """eegclip_wgrad_tok_reduce_adamw == eegclip_wgrad_tok_reduce followed by eegclip_adamw_step_zero_grad over the whole run, BIT FOR BIT: the problems' weight
    and bias gradients lie scattered inside one flat gradient run with uncovered stretches between them (parameters whose gradients came from elsewhere),
    the run's old gradient content is part of the sum, and the gradient run ends up zero."""
    rng = np.random.default_rng(5 + slices)
    names = ["ffn2", "qkv", "out_proj", "nobias"]
    B = 2
    sizes = [(CASES[nm][0], CASES[nm][1], CASES[nm][5]) for nm in names]
    # layout of the run: gap | W0 | b0 | gap | W1 | gap | b1 | W2 | b2 | W3 | gap
    lay, at = [], 37
    for i, (M, N, br) in enumerate(sizes):
        w_off = at
        at += M * N + (11 if i % 2 else 0)
        b_off = at if br else None
        at += (M if br else 0) + (5 if i == 1 else 0)
        lay.append((w_off, b_off))
    n = at + 23
    g0 = rng.standard_normal(n).astype(np.float32) * 0.01
    p0, m0, v0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32) * 0.1, rng.random(n).astype(np.float32) * 0.01
    hyper = (3e-4, 0.9, 0.999, 1e-8, 0.01)

    def run(fused):
        G, P, M_, V = be.dev(g0.copy()), be.dev(p0.copy()), be.dev(m0.copy()), be.dev(v0.copy())
        r2 = np.random.default_rng(11)
        probs = (_abi.WgradTokProblem * len(names))()
        keep = []
        for i, nm in enumerate(names):
            M, N, mg, hm, hn, br = CASES[nm]
            dy = (r2.standard_normal((64 * B, M)) * 0.5).astype(np.float32)
            x = r2.standard_normal((64 * B, N)).astype(np.float32)
            per = M // mg
            a = [make_planes(be, dy[:, g * per:(g + 1) * per], hm, 0) for g in range(mg)]
            if mg > 1:
                a = [be.dev(np.concatenate([be.host(v) for v in a]))]
            b = make_planes(be, x, hn, br == 1)
            keep += [a, b]
            w_off, b_off = lay[i]
            probs[i] = _abi.WgradTokProblem(a=be.ptr(a[0]), b=be.ptr(b), a_group_stride=B * 65536 if mg > 1 else 0, m_groups=mg, heads_m=hm, heads_n=hn, M=M, N=N,
                                            out=be.ptr(G) + 4 * w_off, ldo=N, bias_out=(be.ptr(G) + 4 * b_off) if br else None, bias_mfma=int(br == 2))
        ws = be.dev(np.full(int(be.lib.eegclip_wgrad_tok_workspace_floats(probs, len(names), B, slices)), np.nan, np.float32))
        assert be.lib.eegclip_wgrad_tok(probs, len(names), B, slices, be.ptr(ws), 0, be.stream) == 0
        if fused:
            assert be.lib.eegclip_wgrad_tok_reduce_adamw(probs, len(names), B, slices, be.ptr(ws), be.ptr(P), be.ptr(G), be.ptr(M_), be.ptr(V), n, *hyper, step, be.stream) == 0
        else:
            assert be.lib.eegclip_wgrad_tok_reduce(probs, len(names), B, slices, be.ptr(ws), be.stream) == 0
            assert be.lib.eegclip_adamw_step_zero_grad(be.ptr(P), be.ptr(G), be.ptr(M_), be.ptr(V), n, *hyper, step, 1.0, None, be.stream) == 0
        be.sync()
        if fused:                                             # a problem outside the run / a strided output: refused, nothing enqueued
            assert be.lib.eegclip_wgrad_tok_reduce_adamw(probs, len(names), B, slices, be.ptr(ws), be.ptr(P), be.ptr(G), be.ptr(M_), be.ptr(V), lay[2][0], *hyper, step, be.stream) < 0
            assert be.lib.eegclip_wgrad_tok_reduce_adamw(probs, len(names), B, slices, be.ptr(ws), be.ptr(P), be.ptr(G), be.ptr(M_), be.ptr(V), n, *hyper, 0, be.stream) < 0
        return [be.host(t) for t in (G, P, M_, V)]

    a, b = run(True), run(False)
    assert not a[0].any() and not b[0].any()                  # zero_grad
    assert np.abs(b[1] - p0).max() > 1e-5                     # (something was stepped)
    for x, y, nm in zip(a[1:], b[1:], "PMV"):
        np.testing.assert_array_equal(x, y, err_msg=nm)


def test_wgrad_tok_is_reproducible_and_shape_independent(be):
    """same operands -> bit-identical gradients from both workgroup shapes and from a second run (ordered slab reduction, no atomics)"""
    res = []
    for variant in (0, 1, 2, 3, 0):
        rng = np.random.default_rng(11)
        B, slices = 4, 2
        dy, x = rng.standard_normal((64 * B, 250)).astype(np.float32), rng.standard_normal((64 * B, 250)).astype(np.float32)
        a, b = make_planes(be, dy, 0, 0), make_planes(be, x, 0, 1)
        out, bias = be.zeros((250, 250)), be.zeros(250)
        p = (_abi.WgradTokProblem * 1)(_abi.WgradTokProblem(a=be.ptr(a), b=be.ptr(b), a_group_stride=0, m_groups=1, heads_m=0, heads_n=0, M=250, N=250,
                                                             out=be.ptr(out), ldo=250, bias_out=be.ptr(bias), bias_mfma=0))
        ws = be.zeros(int(be.lib.eegclip_wgrad_tok_workspace_floats(p, 1, B, slices)))
        assert be.lib.eegclip_wgrad_tok(p, 1, B, slices, be.ptr(ws), variant, be.stream) == 0
        assert be.lib.eegclip_wgrad_tok_reduce(p, 1, B, slices, be.ptr(ws), be.stream) == 0
        be.sync()
        res.append((be.host(out).copy(), be.host(bias).copy()))
    for o, bb in res[1:]:
        np.testing.assert_array_equal(o, res[0][0])
        np.testing.assert_array_equal(bb, res[0][1])


def test_wgrad_tok_uneven_slices(be):
    run_problems(be, ["embed", "qkv"], 3, 4, 0, seed=4)                          # 6 k-tiles over 4 slices: 1, 2, 1, 2 per workgroup


def joint_case(be, ids, slices, seed):
    """the joint-subject value embedding (Embed.py:142-144): one problem per subject present, each contracting over that subject's samples only --
    a range of the subject-ordered sample list; the planes stay in batch order"""
    rng = np.random.default_rng(seed)
    ids = np.asarray(ids)
    B = len(ids)
    dy, x = rng.standard_normal((64 * B, 250)).astype(np.float32), rng.standard_normal((64 * B, 250)).astype(np.float32)
    a, b = make_planes(be, dy, 0, 0), make_planes(be, x, 0, 1)
    perm = np.argsort(ids, kind="stable").astype(np.int32)
    PERM = be.dev(perm)
    subjects = sorted(set(ids.tolist()))
    probs = (_abi.WgradTokProblem * len(subjects))()
    outs = []
    start = 0
    for i, sj in enumerate(subjects):
        n = int((ids == sj).sum())
        out, bias = be.zeros((250, 250)), be.zeros(250)
        probs[i] = _abi.WgradTokProblem(a=be.ptr(a), b=be.ptr(b), a_group_stride=0, m_groups=1, heads_m=0, heads_n=0, M=250, N=250, out=be.ptr(out), ldo=250,
                                        bias_out=be.ptr(bias), bias_mfma=0, sample0=start, samples=n, sample_index=be.ptr(PERM))
        rows = np.concatenate([np.arange(64 * sb, 64 * sb + 64) for sb in np.flatnonzero(ids == sj)])
        outs.append((sj, out, bias, reference(dy[rows], x[rows])))
        start += n
    ws = be.dev(np.full(int(be.lib.eegclip_wgrad_tok_workspace_floats(probs, len(subjects), B, slices)), np.nan, np.float32))
    assert be.lib.eegclip_wgrad_tok(probs, len(subjects), B, slices, be.ptr(ws), 0, be.stream) == 0
    assert be.lib.eegclip_wgrad_tok_reduce(probs, len(subjects), B, slices, be.ptr(ws), be.stream) == 0
    be.sync()
    for sj, out, bias, (ew, eb) in outs:
        np.testing.assert_allclose(be.host(out).astype(np.float64), ew, atol=2e-6 * np.abs(ew).max() * np.sqrt(64 * B), err_msg=f"subject {sj}")
        np.testing.assert_allclose(be.host(bias).astype(np.float64), eb, atol=2e-6 * np.abs(eb).max() * np.sqrt(64 * B) + 1e-5, err_msg=f"subject {sj} bias")


@pytest.mark.parametrize("ids,slices", [([3, 0, 3, 7, 0, 3], 2), ([5, 5, 5], 1), ([9, 1, 1, 1, 1, 4], 4)], ids=["mixed", "uniform", "empty-slices"])
def test_wgrad_tok_per_subject_sample_ranges(be, ids, slices):
    joint_case(be, ids, slices, seed=len(ids))


@pytest.mark.gpu
def test_wgrad_tok_per_subject_full_size():
    from backends import get
    rng = np.random.default_rng(2)
    joint_case(get("gpu"), rng.integers(0, 10, 256), 8, seed=9)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_wgrad_tok_full_size(variant):
    from backends import get
    b = get("gpu")
    B = 256
    s = int(b.lib.eegclip_wgrad_tok_slices(3, B))
    assert s % 8 == 0                                         # the XCD-aware workgroup order is in use
    run_problems(b, ["ffn2", "ffn1", "out_proj"], B, s, variant, seed=6)
    run_problems(b, ["qkv"], B, int(b.lib.eegclip_wgrad_tok_slices(3, B)), variant, seed=7)
    run_problems(b, ["embed"], B, int(b.lib.eegclip_wgrad_tok_slices(1, B)), variant, seed=8)


def test_wgrad_tok_rejects_bad_arguments(be):
    p = (_abi.WgradTokProblem * 1)()
    x = be.dev(np.zeros(2 * 32768, np.uint16))
    o = be.zeros((250, 250))
    ws = be.zeros(1 << 20)
    p[0] = _abi.WgradTokProblem(a=be.ptr(x), b=be.ptr(x), a_group_stride=0, m_groups=1, heads_m=0, heads_n=0, M=250, N=256, out=be.ptr(o), ldo=256,
                                bias_out=be.ptr(o), bias_mfma=0)
    assert be.lib.eegclip_wgrad_tok(p, 1, 2, 1, be.ptr(ws), 0, be.stream) != 0     # 256 real columns leave no ones column: bias_mfma is required
    p[0].bias_mfma = 1
    assert be.lib.eegclip_wgrad_tok(p, 1, 2, 8, be.ptr(ws), 0, be.stream) != 0     # more slices than k-tiles
    many = (_abi.WgradTokProblem * 13)(*[p[0]] * 13)
    assert be.lib.eegclip_wgrad_tok(many, 13, 2, 1, be.ptr(ws), 0, be.stream) != 0  # at most 12 problems per launch
    p[0].sample0, p[0].samples = 1, 2
    assert be.lib.eegclip_wgrad_tok(p, 1, 2, 1, be.ptr(ws), 0, be.stream) != 0     # sample range outside the batch
    p[0].sample0, p[0].samples = 0, 0
    p[0].M = 257
    assert be.lib.eegclip_wgrad_tok(p, 1, 2, 1, be.ptr(ws), 0, be.stream) != 0
