"""oracle/prior.py vs the fixtures recorded from the reference's Generation/diffusion_prior.py (rows E1-E3)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN, SEED
from eeg_image_decode_amd import synthetic as syn
from oracle import loops as oloops
from oracle import prior as oprior


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _state():
    return oloops.torch_state(syn.make_state(SEED + 20, oprior.prior_state_spec()))


def test_prior_keys_and_param_count():
    with open(os.path.join(GOLDEN, "prior_keys.json")) as f:
        ref = json.load(f)
    spec = oprior.prior_state_spec()
    assert [k for k, _, _ in spec] == list(ref["keys"].keys())
    assert sum(int(np.prod(s)) for _, s, _ in spec) == ref["n_params"] == 9675648     # matches the notebook's printout too


def test_prior_forward_matches_reference(golden):
    g = golden("prior.npz")
    P = _state()
    x = T(syn.unit_features(SEED + 21, 6, tag="px") * 8.0)
    c = T(syn.unit_features(SEED + 21, 6, tag="pc") * 32.0)
    tt = torch.tensor([0, 5, 333, 999, 20, 980])
    np.testing.assert_allclose(oprior.prior_unet_forward(P, x, tt, c).numpy(), g["eps_cond"], atol=2e-5)
    np.testing.assert_allclose(oprior.prior_unet_forward(P, x, tt).numpy(), g["eps_uncond"], atol=2e-5)
    np.testing.assert_allclose(oprior.prior_unet_forward(P, x, tt.float(), c).numpy(), g["eps_float_t"], atol=2e-5)


def _train_inputs():
    Bn = 32
    h = T(syn.unit_features(SEED + 22, Bn, tag="ph") * 6.0)
    cc = T(syn.unit_features(SEED + 22, Bn, tag="pcc") * 32.0)
    noise = T(syn.eeg_batch(SEED + 22, Bn, 1, 1024)[:, 0])
    ts = torch.from_numpy(np.random.default_rng(SEED + 22).integers(0, 1000, Bn))
    return h, cc, noise, ts


def test_prior_training_objective_and_grads(golden):
    g = golden("prior.npz")
    P = {k: v.requires_grad_(True) for k, v in _state().items()}
    h, cc, noise, ts = _train_inputs()
    loss, pred = oprior.prior_loss(P, h, noise, ts, cc, oprior.DDPMSchedulerOracle())
    loss.backward()
    assert abs(float(loss) - float(g["train_loss"])) < 1e-5
    np.testing.assert_allclose(pred.detach().numpy()[:, :64], g["train_pred_head"], atol=2e-5)
    for k, p in P.items():
        assert abs(float(p.grad.norm()) - float(g["gnorm:" + k])) <= 2e-4 * max(float(g["gnorm:" + k]), 1e-4), k


def test_pipe_train_control_flow(golden):
    g = golden("prior.npz")
    h, cc, _, _ = _train_inputs()
    data = [{"c_embedding": cc[:16], "h_embedding": h[:16]}, {"c_embedding": cc[16:], "h_embedding": h[16:]}]
    P0 = _state()
    torch.manual_seed(1234)
    P1, losses, lrs = oprior.pipe_train(P0, data, 2, 1e-3)
    assert lrs == [1e-3 * k / 500 for k in (1, 2, 3, 4)]          # scheduler stepped BEFORE the optimizer
    dn = float(torch.sqrt(sum(((P1[k] - P0[k]) ** 2).sum() for k in P0)))
    assert abs(dn - float(g["pipe_train_dnorm_total"])) < 2e-3 * float(g["pipe_train_dnorm_total"])
    np.testing.assert_allclose(P1["output_layer.weight"].numpy()[:4, :16], g["pipe_train_out_w_head"], atol=2e-6)


def test_generate_trajectory(golden):
    g = golden("prior.npz")
    P = _state()
    _, cc, _, _ = _train_inputs()
    hf, traj = oprior.generate(P, oprior.DDPMSchedulerOracle(), cc[:1], 50, 5.0, generator=torch.Generator().manual_seed(77))
    assert len(traj) == 50
    np.testing.assert_allclose(hf.numpy(), g["gen_final"], atol=2e-4)
    hu, _ = oprior.generate(P, oprior.DDPMSchedulerOracle(), None, 10, 5.0, generator=torch.Generator().manual_seed(78))
    np.testing.assert_allclose(hu.numpy(), g["gen_final_uncond_10steps"], atol=2e-4)


def test_ddpm_scheduler_identities():
    s = oprior.DDPMSchedulerOracle()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(980, -1, -20))       # "leading" spacing
    x0, n = torch.randn(4, 8), torch.randn(4, 8)
    t = torch.tensor([0, 10, 500, 999])
    xt = s.add_noise(x0, n, t)
    ac = s.alphas_cumprod[t][:, None]
    np.testing.assert_allclose(((xt - (1 - ac).sqrt() * n) / ac.sqrt()).numpy(), x0.numpy(), atol=3e-4)
    # last step (t=0): no noise, prev = posterior mean with alpha_prev = 1 -> the clamped x0 itself
    out = s.step(n, 0, xt).prev_sample
    sa, sb, c0, ct, sig = s.step_coeffs(0)
    assert sig == 0.0 and abs(ct) < 1e-7 and abs(c0 - 1.0) < 1e-6


def test_cosine_warmup_schedule_is_pinned_by_the_transformers_implementation():
    """diffusers' get_cosine_schedule_with_warmup (diffusion_prior.py:287) is a copy of transformers' function of the same name, and
    transformers IS installed here: the oracle's and the product's restatements must reproduce it for every update of a 1200-step run, with
    the reference's order of calls (scheduler.step() before optimizer.step(), :331-332)."""
    from transformers.optimization import get_cosine_schedule_with_warmup
    from eeg_image_decode_amd.prior import cosine_with_warmup_lr as product_lr
    for total in (600, 1200, 400):
        dummy = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
        sch = get_cosine_schedule_with_warmup(dummy, num_warmup_steps=500, num_training_steps=total)
        for step in range(1, total + 1):
            sch.step()
            want = dummy.param_groups[0]["lr"]
            dummy.step()
            assert abs(oprior.cosine_with_warmup_lr(step, 1e-3, 500, total) - want) <= 1e-18 + 1e-12 * want, (total, step)
            assert abs(product_lr(step, 1e-3, 500, total) - want) <= 1e-18 + 1e-12 * want, (total, step)
