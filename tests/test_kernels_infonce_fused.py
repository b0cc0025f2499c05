"""csrc/infonce_fused.hip through the C ABI on both backends: row log-sum-exp / loss / positives of a logits block that is never written,
the gradient tile G (one or two normalisers) and d loss / d scale, in both arithmetic modes and both tile sizes, several blocks per launch."""
import ctypes

import numpy as np
import pytest

from backends import be  # noqa: F401
from eeg_image_decode_amd import _abi
from test_kernels_gemm_x3 import bf16_round, split


def planes_arg(planes, tile, waves=0):
    """planes | tile size << 8 (0 = the library's choice, 255 = 256 x 256) | waves per workgroup << 16 (1 = 4 waves, 2 = 8 waves: 128-tiles only, 3 = 4 MFMA waves + 4 producer waves, 0 = the library's choice)"""
    return planes | (tile << 8) | (waves << 16)


def feats(rng, rows, D, kind):
    x = rng.standard_normal((rows, D)).astype(np.float32)
    if kind == "unit":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    else:                                           # LayerNorm output: zero mean, unit variance per row
        x = (x - x.mean(1, keepdims=True)) / x.std(1, keepdims=True)
    return x.astype(np.float32)


def prep(be, x, planes):
    X = be.dev(x)
    hi, lo = be.zeros(x.shape, np.uint16), (be.zeros(x.shape, np.uint16) if planes == 2 else None)
    assert be.lib.eegclip_split_bf16(be.ptr(X), be.ptr(hi), be.ptr(lo), x.size, be.stream) == 0
    return X, hi, lo


def logits_ref(q, k, planes, s):
    if planes == 1:
        return s * (bf16_round(q).astype(np.float64) @ bf16_round(k).astype(np.float64).T)
    qh, ql = split(q)
    kh, kl = split(k)
    return s * (qh @ kh.T + qh @ kl.T + ql @ kh.T)


def lse(x):
    m = x.max(1, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(1, keepdims=True)))[:, 0]


@pytest.mark.parametrize("planes", [1, 2])
@pytest.mark.parametrize("n,N,D,tile,col0,waves", [(64, 64, 64, 64, 0, 0), (64, 192, 128, 64, 128, 0), (128, 256, 64, 128, 64, 1), (128, 256, 64, 128, 64, 2),
                                                   (128, 384, 128, 128, 256, 2), (128, 384, 128, 128, 256, 3), (128, 128, 192, 64, 0, 0), (64, 192, 128, 64, 128, 1),
                                                   (256, 512, 128, 255, 256, 0), (256, 256, 64, 255, 0, 0)])   # tile code 255 = 256 x 256 (one product only; with two planes the library falls back to its own choice)
def test_fused_forward_and_gradient_blocks(be, planes, n, N, D, tile, col0, waves):
    rng = np.random.default_rng(n + N + D + planes)
    s = 2.6593
    blocks = [(feats(rng, n, D, "ln"), feats(rng, N, D, "unit"), col0, 0.495), (feats(rng, n, D, "unit") * 3, feats(rng, N, D, "ln"), col0, 0.005)]
    L = be.lib
    SC, LOSS, DS = be.dev(np.array([s], np.float32)), be.zeros(1), be.zeros(1)
    ws = int(L.eegclip_infonce_fused_workspace_floats(n, N))
    keep, probs, refs = [], [], []
    for q, k, c0, w in blocks:
        Q, qh, ql = prep(be, q, planes)
        K, kh, kl = prep(be, k, planes)
        part, diag, lse_o, G = be.zeros(ws), be.zeros(n), be.zeros(n), be.dev(np.full((n, N), np.nan, np.float32))
        keep.append((Q, qh, ql, K, kh, kl, part, diag, lse_o, G))
        probs.append(_abi.InfonceProblem(q_hi=be.ptr(qh), q_lo=be.ptr(ql), k_hi=be.ptr(kh), k_lo=be.ptr(kl), col0=c0, weight=w, part=be.ptr(part),
                                         diag=be.ptr(diag), lse=be.ptr(lse_o), lse_k=None, G=be.ptr(G), ldg=N))
        refs.append(logits_ref(q, k, planes, s))
    arr = (_abi.InfonceProblem * len(probs))(*probs)
    n_total = 2 * n
    assert L.eegclip_infonce_fused_fwd(arr, len(probs), n, N, D, planes_arg(planes, tile, waves), n_total, be.ptr(SC), be.ptr(LOSS), be.stream) == 0
    want_loss = 0.0
    for (q, k, c0, w), S, kp in zip(blocks, refs, keep):
        l_ref = lse(S)
        pos = S[np.arange(n), c0 + np.arange(n)]
        np.testing.assert_allclose(be.host(kp[8]), l_ref, atol=3e-5)
        np.testing.assert_allclose(be.host(kp[7]), pos, atol=3e-5)
        want_loss += w / n_total * (l_ref - pos).sum()
    assert abs(float(be.host(LOSS)[0]) - want_loss) < 2e-5 * max(1.0, abs(want_loss))
    # exact-product check of the parity arithmetic: the split logits are within 1e-4 of fp64 products (budget 1e-3)
    if planes == 2:
        q, k = blocks[0][0], blocks[0][1]
        assert np.abs(refs[0] - s * q.astype(np.float64) @ k.astype(np.float64).T).max() < 1e-4
    # gradient tiles: one normaliser (row-sharded blocks) ...
    assert L.eegclip_infonce_fused_grad(arr, len(probs), n, N, D, planes_arg(planes, tile, waves), n_total, be.ptr(SC), be.ptr(DS), be.stream) == 0
    want_ds = 0.0
    for (q, k, c0, w), S, kp in zip(blocks, refs, keep):
        Pm = np.exp(S - lse(S)[:, None])
        Pm[np.arange(n), c0 + np.arange(n)] -= 1.0
        Gp = w / n_total * Pm
        np.testing.assert_allclose(be.host(kp[9]), s * Gp, atol=2e-6)
        want_ds += (Gp * (S / s)).sum()
    assert abs(float(be.host(DS)[0]) - want_ds) < 2e-5 * max(1.0, abs(want_ds))


@pytest.mark.parametrize("planes,tile,waves", [(1, 64, 0), (2, 64, 1), (2, 128, 1), (2, 128, 2), (1, 128, 2), (1, 128, 3), (2, 128, 0), (1, 255, 0)])
def test_fused_symmetric_square_case_one_gradient_tile_for_both_terms(be, planes, tile, waves):
    """single-process ClipLoss: blocks (A, B) and (B, A); the gradient w.r.t. A needs G = c (P_row + P_col - 2 I): the row normaliser of
    the block and the row normaliser of the swapped block as `lse_k`"""
    rng = np.random.default_rng(5 + planes)
    n, D, s = (256 if tile == 255 else 128), 64, 2.6593          # (tile code 255: one 256 x 256 tile, one partial slot per row and key tile)
    a, b = feats(rng, n, D, "ln"), feats(rng, n, D, "unit")
    L = be.lib
    SC, LOSS, DS = be.dev(np.array([s], np.float32)), be.zeros(1), be.zeros(1)
    A, ah, al = prep(be, a, planes)
    B, bh, bl = prep(be, b, planes)
    ws = int(L.eegclip_infonce_fused_workspace_floats(n, n))
    bufs = [(be.zeros(ws), be.zeros(n), be.zeros(n)) for _ in range(2)]
    G = be.zeros((n, n))
    p_ab = _abi.InfonceProblem(q_hi=be.ptr(ah), q_lo=be.ptr(al), k_hi=be.ptr(bh), k_lo=be.ptr(bl), col0=0, weight=0.5, part=be.ptr(bufs[0][0]),
                               diag=be.ptr(bufs[0][1]), lse=be.ptr(bufs[0][2]), lse_k=be.ptr(bufs[1][2]), G=be.ptr(G), ldg=n)
    p_ba = _abi.InfonceProblem(q_hi=be.ptr(bh), q_lo=be.ptr(bl), k_hi=be.ptr(ah), k_lo=be.ptr(al), col0=0, weight=0.5, part=be.ptr(bufs[1][0]),
                               diag=be.ptr(bufs[1][1]), lse=be.ptr(bufs[1][2]), lse_k=None, G=None, ldg=0)
    arr = (_abi.InfonceProblem * 2)(p_ab, p_ba)
    assert L.eegclip_infonce_fused_fwd(arr, 2, n, n, D, planes_arg(planes, tile, waves), n, be.ptr(SC), be.ptr(LOSS), be.stream) == 0
    one = (_abi.InfonceProblem * 1)(p_ab)
    assert L.eegclip_infonce_fused_grad(one, 1, n, n, D, planes_arg(planes, tile, waves), n, be.ptr(SC), be.ptr(DS), be.stream) == 0
    S = logits_ref(a, b, planes, s)
    lr, lc = lse(S), lse(S.T)
    want = 0.5 / n * ((lr - np.diag(S)).sum() + (lc - np.diag(S)).sum())           # models/loss.py:136-139
    assert abs(float(be.host(LOSS)[0]) - want) < 2e-5 * max(1.0, abs(want))
    Gp = 0.5 / n * (np.exp(S - lr[:, None]) + np.exp(S - lc[None, :]) - 2 * np.eye(n))
    np.testing.assert_allclose(be.host(G), s * Gp, atol=2e-6)
    assert abs(float(be.host(DS)[0]) - (Gp * S / s).sum()) < 2e-5
    # training form: the forward leaves only the partials (loss = NULL) and the gradient pass finalises them itself -- same G, d scale and loss, no
    # finalize launch; lse / lse_k are not read (poisoned here)
    LOSS2, DS2, G2 = be.zeros(1), be.zeros(1), be.dev(np.full((n, n), np.nan, np.float32))
    POISON = be.dev(np.full(n, np.nan, np.float32))
    assert L.eegclip_infonce_fused_fwd(arr, 2, n, n, D, planes_arg(planes, tile, waves), n, be.ptr(SC), None, be.stream) == 0
    p_fin = _abi.InfonceProblem(q_hi=be.ptr(ah), q_lo=be.ptr(al), k_hi=be.ptr(bh), k_lo=be.ptr(bl), col0=0, weight=0.5, part=be.ptr(bufs[0][0]),
                                diag=be.ptr(bufs[0][1]), lse=be.ptr(POISON), lse_k=be.ptr(POISON), G=be.ptr(G2), ldg=n, part_k=be.ptr(bufs[1][0]),
                                diag_k=be.ptr(bufs[1][1]))
    fin = (_abi.InfonceProblem * 1)(p_fin)
    assert L.eegclip_infonce_fused_grad_finalize(fin, 1, n, n, D, planes_arg(planes, tile, waves), n, be.ptr(SC), be.ptr(LOSS2), be.ptr(DS2), be.stream) == 0
    np.testing.assert_allclose(be.host(G2), be.host(G), rtol=2e-5, atol=1e-9)
    assert abs(float(be.host(LOSS2)[0]) - want) < 2e-5 * max(1.0, abs(want))
    assert abs(float(be.host(DS2)[0]) - float(be.host(DS)[0])) < 2e-5
    # argument checks: the plain gradient entry refuses the folded form and vice versa, rectangular blocks are refused
    assert L.eegclip_infonce_fused_grad(fin, 1, n, n, D, planes_arg(planes, tile, waves), n, be.ptr(SC), be.ptr(DS2), be.stream) < 0
    assert L.eegclip_infonce_fused_grad_finalize(one, 1, n, n, D, planes_arg(planes, tile, waves), n, be.ptr(SC), be.ptr(LOSS2), be.ptr(DS2), be.stream) < 0
    assert L.eegclip_infonce_fused_grad_finalize(fin, 1, n, 2 * n, D, planes_arg(planes, tile, waves), n, be.ptr(SC), be.ptr(LOSS2), be.ptr(DS2), be.stream) < 0


def test_fused_rejects_unsupported_shapes(be):
    L = be.lib
    assert L.eegclip_infonce_fused_supported(256, 2048, 1024) == 1
    assert L.eegclip_infonce_fused_supported(200, 200, 1024) == 0 and L.eegclip_infonce_fused_supported(64, 64, 32) == 0
    z = be.zeros(64)
    p = (_abi.InfonceProblem * 1)(_abi.InfonceProblem(q_hi=be.ptr(z), q_lo=None, k_hi=be.ptr(z), k_lo=None, col0=0, weight=1.0, part=be.ptr(z), diag=be.ptr(z),
                                                      lse=be.ptr(z), lse_k=None, G=None, ldg=0))
    assert L.eegclip_infonce_fused_fwd(p, 1, 200, 200, 1024, 1, 200, be.ptr(z), be.ptr(z), be.stream) < 0
    assert L.eegclip_infonce_fused_fwd(p, 1, 64, 64, 64, 2, 64, be.ptr(z), be.ptr(z), be.stream) < 0        # planes = 2 without lo planes
    assert L.eegclip_infonce_fused_fwd(p, 9, 64, 64, 64, 1, 64, be.ptr(z), be.ptr(z), be.stream) < 0
