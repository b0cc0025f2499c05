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
    monkeypatch.setenv("EEGCLIP_TOKEN_BLOCK_BWD", "0")       # (with the fused backward the forward does not store n2; here every tensor is compared)
    m = _model(dev)
    m.train(train)
    x = torch.from_numpy(syn.eeg_batch(SEED + 31, B)).to(dev)
    torch.manual_seed(1234)                                    # the engine draws the step's Philox seed from torch's generator
    with torch.no_grad():
        z = m(x, subject)
    eng = m._engine()
    names = eng.plans[next(k for k in eng.plans if k[0] == "f")].op_names()
    assert ("eegclip_token_block_fwd" in names) == fused and ("eegclip_attention_fwd" in names) == (not fused)
    return z.cpu().numpy(), {k: eng.bufs[B][k].detach().cpu().numpy().copy() for k in SAVED}


def check_fused_forward_equals_the_unfused_plan(dev, B, train, subject, monkeypatch):
    z0, s0 = _forward(dev, B, train, subject, False, monkeypatch)
    z1, s1 = _forward(dev, B, train, subject, True, monkeypatch)
    for k in SAVED:
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


def _grads(dev, B, train, subject, fused, monkeypatch, wgrad_tr=False):
    monkeypatch.setenv("EEGCLIP_TOKEN_BLOCK", "1" if fused else "0")
    monkeypatch.delenv("EEGCLIP_TOKEN_BLOCK_BWD", raising=False)
    monkeypatch.setenv("EEGCLIP_WGRAD_TR", "1" if wgrad_tr else "0")
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
    assert ("eegclip_wgrad_tr" in names) == (fused and wgrad_tr)
    act = {k: eng.bufs[B][k].detach().cpu().numpy().copy() for k in ("df2", "dg1", "da1", "dctx", "dqkv", "dr1")}
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


def check_weight_gradients_from_natural_planes(dev, B, monkeypatch):
    """EEGCLIP_WGRAD_TR=1: the block's weight gradients through eegclip_split_rows_natural + eegclip_wgrad_tr against the plan GEMMs"""
    g0, _ = _grads(dev, B, True, 1, True, monkeypatch)
    g1, _ = _grads(dev, B, True, 1, True, monkeypatch, wgrad_tr=True)
    assert g0.keys() == g1.keys()
    for k, r in g0.items():
        if k.endswith("key_projection.bias"):
            continue
        np.testing.assert_allclose(g1[k], r, atol=1e-3 * float(np.abs(r).max()) + 1e-7, err_msg=k)


@pytest.mark.emu
def test_weight_gradients_from_natural_planes_on_the_emulator(monkeypatch):
    from emu_patch import product_on_emulator
    with product_on_emulator():
        check_weight_gradients_from_natural_planes("cpu", 2, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [3, 256])
def test_weight_gradients_from_natural_planes_on_the_gpu(B, monkeypatch):
    check_weight_gradients_from_natural_planes("cuda", B, monkeypatch)
