"""csrc/token_block.hip -- the encoder's transformer block as one workgroup per sample -- against the launch-per-Linear forward plan it replaces
(EEGCLIP_TOKEN_BLOCK=0), tensor by tensor: every activation the backward reads, with real dropout (the same Philox masks by construction: same
seed, sites and flat element indices), for the subject-token table and the shared token, then the whole training step against the oracle.  The
unfused plan is itself pinned by the reference fixtures (tests/test_model_gpu.py); on the emulator the same body runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from conftest import SEED
from eeg_image_decode_amd import synthetic as syn
from oracle import atms as oatms

SAVED = ["h", "qkv", "ctx", "r1", "n1", "mu1", "rs1", "f1", "g1", "r2", "n2", "mu2", "rs2", "n3", "mu3", "rs3"]


def _model(dev):
    from eeg_image_decode_amd.atms import ATMS
    state_np = syn.make_state(SEED, oatms.state_spec())
    m = ATMS()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state_np.items()})
    return m.to(dev)


def _forward(dev, B, train, subject, fused, monkeypatch):
    monkeypatch.setenv("EEGCLIP_TOKEN_BLOCK", "1" if fused else "0")
    m = _model(dev)
    m.train(train)
    x = torch.from_numpy(syn.eeg_batch(SEED + 31, B)).to(dev)
    torch.manual_seed(1234)                                    # the engine draws the step's Philox seed from torch's generator
    with torch.no_grad():
        z = m(x, subject)
    eng = m._engine()
    names = eng.plans[next(k for k in eng.plans if k[0] == "f")].op_names()
    assert ("eegclip_token_block_fwd" in names) == fused and ("eegclip_attention_fwd" in names) == (not fused)
    # (the fused kernel leaves ctx / n1 / g1 -- read only by the weight-gradient GEMMs -- as token planes and does not store n2: saved_f32 rebuilds fp32)
    saved = {k: eng.saved_f32(B, k).detach().cpu().numpy().copy() for k in SAVED if not (fused and k == "n2")}
    if fused:
        xp = eng.saved_f32(B, "x").detach().cpu().numpy().reshape(B, 64, 250)      # the EEG sample as planes: token row 1 + channel, row 0 zero
        np.testing.assert_array_equal(xp[:, 0], 0.0)
        np.testing.assert_allclose(xp[:, 1:], x.cpu().numpy(), rtol=2.0 ** -16, atol=1e-30)
        ones = {k: eng.bufs[B][k + "p"].to(torch.float32)[:, 0, :, 255].cpu().numpy() for k in ("x", "h", "ctx", "n1")}      # the bias-gradient column
        assert all((v[:, 1:] == 1.0).all() for v in ones.values()) and (ones["x"][:, 0] == 0.0).all() and (ones["h"][:, 0] == 1.0).all()
    return z.cpu().numpy(), saved


def check_fused_forward_equals_the_unfused_plan(dev, B, train, subject, monkeypatch):
    z0, s0 = _forward(dev, B, train, subject, False, monkeypatch)
    z1, s1 = _forward(dev, B, train, subject, True, monkeypatch)
    for k in s1:
        a, r = s1[k].reshape(-1), s0[k].reshape(-1)
        if k == "h":                                           # (row 63 of h / qkv / ... exists in both: token 63 = EEG channel 62)
            assert (a == 0).mean() == pytest.approx((r == 0).mean(), abs=1e-9) or not train      # identical dropout pattern
        np.testing.assert_allclose(a, r, atol=3e-4 * max(1.0, float(np.abs(r).max())), err_msg=k)
        assert np.abs(a - r).mean() < 2e-5 * max(1.0, float(np.abs(r).mean())), k
    np.testing.assert_allclose(z1, z0, atol=2e-4)


@pytest.mark.emu
@pytest.mark.parametrize("train,subject", [(True, 1), (False, 10), (True, None)])
def test_fused_token_block_equals_the_unfused_plan_on_the_emulator(train, subject, monkeypatch):
    from emu_patch import product_on_emulator
    with product_on_emulator():
        check_fused_forward_equals_the_unfused_plan("cpu", 2, train, subject, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [3, 256])
@pytest.mark.parametrize("train,subject", [(True, 1), (False, 10), (True, None)])
def test_fused_token_block_equals_the_unfused_plan_on_the_gpu(B, train, subject, monkeypatch):
    check_fused_forward_equals_the_unfused_plan("cuda", B, train, subject, monkeypatch)


def _grads(dev, B, train, subject, fused, monkeypatch, variant=0):
    monkeypatch.setenv("EEGCLIP_TOKEN_BLOCK", "1" if fused else "0")
    monkeypatch.setenv("EEGCLIP_WGRAD_VARIANT", str(variant))
    m = _model(dev)
    m.train(train)
    x = torch.from_numpy(syn.eeg_batch(SEED + 32, B)).to(dev)
    tgt = torch.from_numpy(syn.unit_features(SEED + 32, B, tag="img")).to(dev)
    torch.manual_seed(4321)
    z = m(x, subject)
    (z * tgt).sum().backward()                                 # a generic upstream gradient (the loss kernels have their own tests)
    eng = m._engine()
    names = eng.plans[next(k for k in eng.plans if k[0] == "b")].op_names()
    assert ("eegclip_token_block_bwd" in names) == fused
    # (one launch for all five weight gradients of the block)
    assert (names.count("eegclip_wgrad_tok") == 1) == fused and ("eegclip_gemm_f32" in names[names.index("eegclip_cstack_bwd_apply" if "eegclip_cstack_bwd_apply" in names else "eegclip_tsconv_bwd_x"):]) == (not fused)
    act = {k: eng.saved_f32(B, k).detach().cpu().numpy().copy() for k in ("df2", "dg1", "da1", "dctx", "dqkv", "dr1")}
    if fused:                                                  # dh as planes (the value embedding's dY) = the fp32 dr1 the same kernel wrote
        np.testing.assert_allclose(eng.saved_f32(B, "dr1").cpu().numpy(), eng.bufs[B]["dr1"].reshape(B * 64, 250).cpu().numpy(), rtol=2.0 ** -16, atol=1e-30)
        act["dr1"] = eng.bufs[B]["dr1"].detach().cpu().numpy().copy()
    return {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}, act


def check_fused_backward_equals_the_unfused_plan(dev, B, train, subject, monkeypatch):
    g0, a0 = _grads(dev, B, train, subject, False, monkeypatch)
    g1, a1 = _grads(dev, B, train, subject, True, monkeypatch)
    for k in a0:                                               # what the weight-gradient GEMMs read
        r = a0[k]
        np.testing.assert_allclose(a1[k], r, atol=5e-4 * max(1e-6, float(np.abs(r).max())), err_msg=k)
    assert g0.keys() == g1.keys()
    for k, r in g0.items():
        if k.endswith("key_projection.bias"):                  # exactly zero in exact arithmetic (softmax shift invariance): round-off in both
            assert np.abs(g1[k]).max() < 1e-5
            continue
        np.testing.assert_allclose(g1[k], r, atol=1e-3 * float(np.abs(r).max()) + 1e-7, err_msg=k)


@pytest.mark.emu
@pytest.mark.parametrize("train,subject", [(True, 1), (False, 10)])
def test_fused_token_block_backward_equals_the_unfused_plan_on_the_emulator(train, subject, monkeypatch):
    from emu_patch import product_on_emulator
    with product_on_emulator():
        check_fused_backward_equals_the_unfused_plan("cpu", 2, train, subject, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [3, 256])
@pytest.mark.parametrize("train,subject", [(True, 1), (False, 10), (True, None)])
def test_fused_token_block_backward_equals_the_unfused_plan_on_the_gpu(B, train, subject, monkeypatch):
    check_fused_backward_equals_the_unfused_plan("cuda", B, train, subject, monkeypatch)


def check_both_workgroup_shapes_of_the_weight_gradient_kernel(dev, B, monkeypatch):
    """EEGCLIP_WGRAD_VARIANT=1 (256-thread workgroups of csrc/wgrad_tok.hip) in the plan against the default (bit-identity of the two kernel shapes on the
    same operands is tests/test_kernels_wgrad.py; here the operands carry the run-to-run round-off of the atomics upstream)"""
    g0, _ = _grads(dev, B, True, 1, True, monkeypatch)
    g1, _ = _grads(dev, B, True, 1, True, monkeypatch, variant=1)
    assert g0.keys() == g1.keys()
    for k, r in g0.items():
        if not k.endswith("key_projection.bias"):
            np.testing.assert_allclose(g1[k], r, atol=2e-4 * float(np.abs(r).max()) + 1e-7, err_msg=k)


@pytest.mark.emu
def test_both_workgroup_shapes_of_the_weight_gradient_kernel_on_the_emulator(monkeypatch):
    from emu_patch import product_on_emulator
    with product_on_emulator():
        check_both_workgroup_shapes_of_the_weight_gradient_kernel("cpu", 2, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [3, 256])
def test_both_workgroup_shapes_of_the_weight_gradient_kernel_on_the_gpu(B, monkeypatch):
    check_both_workgroup_shapes_of_the_weight_gradient_kernel("cuda", B, monkeypatch)


def check_stats_tail_equals_the_statistics_launch(dev, B, monkeypatch):
    """(round 6) the conv stack's BatchNorm1 batch sums as the TAIL of the block kernel (eegclip_token_block_desc.cs_rows) against the eegclip_cstack_stats1
    launch it replaces (EEGCLIP_STATS1_TAIL=0): the same device function over the same n3 rows -- the partial rows, and with them the embeddings and the
    BatchNorm running statistics, are identical bit for bit."""
    res = []
    for tail in ("1", "0"):
        monkeypatch.setenv("EEGCLIP_STATS1_TAIL", tail)
        monkeypatch.setenv("EEGCLIP_TOKEN_BLOCK", "1")
        m = _model(dev).train()
        x = torch.from_numpy(syn.eeg_batch(SEED + 33, B)).to(dev)
        torch.manual_seed(77)
        with torch.no_grad():
            z = m(x, 1)
        eng = m._engine()
        names = eng.plans[next(k for k in eng.plans if k[0] == "f")].op_names()
        assert ("eegclip_cstack_stats1" in names) == (tail == "0") and "eegclip_cstack_fwd" in names
        res.append((eng.bufs[B]["cs_rows"][0].cpu().numpy().copy(), z.cpu().numpy(), {k: v.cpu().numpy().copy() for k, v in m.state_dict().items() if "running" in k}))
    (r1, z1, bn1), (r0, z0, bn0) = res
    assert np.isfinite(r1).all() and np.abs(r1).max() > 0
    np.testing.assert_array_equal(r1, r0)
    np.testing.assert_array_equal(z1, z0)
    for k in bn1:
        np.testing.assert_array_equal(bn1[k], bn0[k], err_msg=k)


@pytest.mark.emu
def test_conv_stack_statistics_as_the_block_kernels_tail_on_the_emulator(monkeypatch):
    from emu_patch import product_on_emulator
    with product_on_emulator():
        check_stats_tail_equals_the_statistics_launch(torch.device("cpu"), 3, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [3, 256])
def test_conv_stack_statistics_as_the_block_kernels_tail_on_the_gpu(B, monkeypatch):
    check_stats_tail_equals_the_statistics_launch(torch.device("cuda"), B, monkeypatch)
