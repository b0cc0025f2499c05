"""MI355X parity of the diffusion prior (rows E1-E3) vs the fixtures recorded from the reference's
Generation/diffusion_prior.py and vs the CPU oracle.  fp32 end to end: tolerances are round-off class."""
import numpy as np
import pytest
import torch

from conftest import SEED
from eeg_image_decode_amd import synthetic as syn
from oracle import loops as oloops
from oracle import prior as oprior

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_prior(dropout=0.0):
    from eeg_image_decode_amd.prior import DiffusionPriorUNet
    m = DiffusionPriorUNet(cond_dim=1024, dropout=dropout)
    state = syn.make_state(SEED + 20, oprior.prior_state_spec())
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    return m.cuda(), oloops.torch_state(state)


def test_prior_state_dict_keys():
    from eeg_image_decode_amd.prior import DiffusionPriorUNet
    m = DiffusionPriorUNet(cond_dim=1024, dropout=0.1)
    assert [k for k, _, _ in oprior.prior_state_spec()] == list(m.state_dict().keys())
    assert sum(p.numel() for p in m.parameters()) == 9675648


def test_prior_forward_matches_reference_fixture(golden):
    g = golden("prior.npz")
    m, _ = make_prior()
    m.eval()
    x = T(syn.unit_features(SEED + 21, 6, tag="px") * 8.0).cuda()
    c = T(syn.unit_features(SEED + 21, 6, tag="pc") * 32.0).cuda()
    tt = torch.tensor([0, 5, 333, 999, 20, 980]).cuda()
    with torch.no_grad():
        np.testing.assert_allclose(m(x, tt, c).cpu().numpy(), g["eps_cond"], atol=2e-4)
        np.testing.assert_allclose(m(x, tt).cpu().numpy(), g["eps_uncond"], atol=2e-4)
        np.testing.assert_allclose(m(x, tt.float(), c).cpu().numpy(), g["eps_float_t"], atol=2e-4)


def _train_inputs():
    Bn = 32
    h = T(syn.unit_features(SEED + 22, Bn, tag="ph") * 6.0)
    cc = T(syn.unit_features(SEED + 22, Bn, tag="pcc") * 32.0)
    noise = T(syn.eeg_batch(SEED + 22, Bn, 1, 1024)[:, 0])
    ts = torch.from_numpy(np.random.default_rng(SEED + 22).integers(0, 1000, Bn))
    return h, cc, noise, ts


def test_prior_objective_and_grads_match_reference_fixture(golden):
    from eeg_image_decode_amd.prior import DDPMScheduler
    g = golden("prior.npz")
    m, P = make_prior()
    m.train()
    h, cc, noise, ts = _train_inputs()
    sched = DDPMScheduler()
    pert = sched.add_noise(h.cuda(), noise.cuda(), ts.cuda())
    pred = m(pert, ts.cuda(), cc.cuda())
    loss = ((pred - noise.cuda()) ** 2).mean()
    loss.backward()
    assert abs(float(loss) - float(g["train_loss"])) < 1e-4
    np.testing.assert_allclose(pred.detach().cpu().numpy()[:, :64], g["train_pred_head"], atol=2e-4)
    for k, p in m.named_parameters():
        ref = float(g["gnorm:" + k])
        assert abs(float(p.grad.norm()) - ref) <= 3e-3 * max(ref, 1e-4), (k, float(p.grad.norm()), ref)
    # full gradients vs the oracle's autograd, with dropout masks shared through Philox
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lo, _ = oprior.prior_loss(Pg, h, noise, ts, cc, oprior.DDPMSchedulerOracle())
    lo.backward()
    for k, p in m.named_parameters():
        r = Pg[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, atol=1e-7 + 3e-3 * np.abs(r).max(), err_msg=k)


def test_prior_train_mode_dropout_matches_oracle_masks():
    from philox_np import keep_mask
    m, P = make_prior(dropout=0.1)
    m.train()
    N = 16
    x = T(syn.unit_features(SEED + 30, N, tag="dx") * 8.0)
    c = T(syn.unit_features(SEED + 30, N, tag="dc") * 32.0)
    tt = torch.from_numpy(np.random.default_rng(5).integers(0, 1000, N))
    out = m(x.cuda(), tt.cuda(), c.cuda())
    out.square().mean().backward()
    seed = m._engine().bufs[N]["seed"]
    hd = [1024, 512, 256, 128, 64]
    outs = [hd[i + 1] for i in range(4)] + [hd[i - 1] for i in range(4, 0, -1)]
    masks = {s: T(keep_mask(seed, s, N * outs[s], 0.1).reshape(N, outs[s])) for s in range(8)}
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    oo = oprior.prior_unet_forward(Pg, x, tt, c, p_drop=0.1, train=True, masks=masks)
    oo.square().mean().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), oo.detach().numpy(), atol=3e-4)
    for k, p in m.named_parameters():
        r = Pg[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, atol=1e-7 + 4e-3 * np.abs(r).max(), err_msg=k)


def test_pipe_train_matches_reference_fixture(golden):
    """E2: Pipe.train control flow -- the torch RNG is consumed on the CPU in the reference's order for this test by running the
    oracle's pipe_train on the same seed and comparing the product against it AND the reference fixture."""
    from eeg_image_decode_amd.prior import Pipe
    g = golden("prior.npz")
    h, cc, _, _ = _train_inputs()
    data = [{"c_embedding": cc[:16], "h_embedding": h[:16]}, {"c_embedding": cc[16:], "h_embedding": h[16:]}]
    m, P0 = make_prior()
    pipe = Pipe(m, device="cuda")
    # inject the reference's CPU noise stream: same seed, same call order (rand(1), randn_like, randint)
    import eeg_image_decode_amd.prior as pp
    real_randn_like, real_randint = torch.randn_like, torch.randint
    try:
        pp.torch.randn_like = lambda t_: real_randn_like(t_.cpu()).to(t_.device)
        pp.torch.randint = lambda lo, hi, shape, device=None: real_randint(lo, hi, shape).to(device)
        torch.manual_seed(1234)
        pipe.train(data, num_epochs=2, learning_rate=1e-3)
    finally:
        pp.torch.randn_like, pp.torch.randint = real_randn_like, real_randint
    assert pipe.lr_history == [1e-3 * k / 500 for k in (1, 2, 3, 4)]
    sd = {k: v.detach().cpu() for k, v in m.named_parameters()}
    dn = float(torch.sqrt(sum(((sd[k] - P0[k]) ** 2).sum() for k in P0)))
    assert abs(dn - float(g["pipe_train_dnorm_total"])) < 5e-3 * float(g["pipe_train_dnorm_total"])
    np.testing.assert_allclose(sd["output_layer.weight"].numpy()[:4, :16], g["pipe_train_out_w_head"], atol=1e-5)


def test_generate_matches_reference_fixture_and_batches_are_independent(golden):
    from eeg_image_decode_amd.prior import Pipe
    g = golden("prior.npz")
    m, _ = make_prior()
    _, cc, _, _ = _train_inputs()
    pipe = Pipe(m, device="cuda")
    hf = pipe.generate(c_embeds=cc[:1], num_inference_steps=50, guidance_scale=5.0, generator=torch.Generator().manual_seed(77))
    np.testing.assert_allclose(hf.cpu().numpy(), g["gen_final"], atol=1e-2)          # north_star: denoised latents within 1e-2
    assert np.abs(hf.cpu().numpy() - g["gen_final"]).max() < 2e-3                      # fp32 path is far tighter
    hu = pipe.generate(c_embeds=None, num_inference_steps=10, guidance_scale=5.0, generator=torch.Generator().manual_seed(78))
    np.testing.assert_allclose(hu.cpu().numpy(), g["gen_final_uncond_10steps"], atol=2e-3)
    # batched sampling == independent chains: feed the batch the concatenation of per-chain noise streams
    class CatGen:
        """generator stand-in is not possible; instead compare N=1 chains run one by one on disjoint conditions"""
    outs = [pipe.generate(c_embeds=cc[i:i + 1], num_inference_steps=10, guidance_scale=5.0, generator=torch.Generator().manual_seed(5)) for i in range(3)]
    # same noise seed for each chain; a batch of 3 with per-row identical noise must reproduce them
    import eeg_image_decode_amd.prior as pp
    real_randn = torch.randn
    try:
        def rep_randn(*shape, generator=None, device=None, dtype=None):
            shp = shape[0] if isinstance(shape[0], (tuple, torch.Size)) else shape
            one = real_randn(1, *shp[1:], generator=generator, device=device, dtype=dtype)
            return one.repeat(shp[0], *([1] * (len(shp) - 1)))
        pp.torch.randn = rep_randn
        hb = pipe.generate(c_embeds=cc[:3], num_inference_steps=10, guidance_scale=5.0, generator=torch.Generator().manual_seed(5))
    finally:
        pp.torch.randn = real_randn
    np.testing.assert_allclose(hb.cpu().numpy(), torch.cat(outs).cpu().numpy(), atol=2e-4)


def test_generate_graph_replay_equals_eager_chain(monkeypatch):
    """Pipe.generate replays the whole DDPM chain from one captured HIP graph; the launch-by-launch path (EEGCLIP_PRIOR_GRAPH=0) must give
    the same latents, a replay must follow new inputs (condition, start latent, noise) and new WEIGHTS without re-capture."""
    from eeg_image_decode_amd.prior import Pipe
    m, _ = make_prior()
    _, cc, _, _ = _train_inputs()
    pipe = Pipe(m, device="cuda")
    run = lambda c, seed: pipe.generate(c_embeds=c, num_inference_steps=12, guidance_scale=5.0, generator=torch.Generator().manual_seed(seed))
    a1, a2 = run(cc[:4], 1), run(cc[4:8], 2)                   # capture + replay, then replay with other inputs
    assert len(pipe._graphs) == 1
    with torch.no_grad():
        m.output_layer.bias.add_(0.25)                          # parameters live in the flat buffer the graph reads
    a3 = run(cc[:4], 1)
    monkeypatch.setenv("EEGCLIP_PRIOR_GRAPH", "0")
    e3 = run(cc[:4], 1)
    with torch.no_grad():
        m.output_layer.bias.sub_(0.25)
    e1, e2 = run(cc[:4], 1), run(cc[4:8], 2)
    # not bit-equal: the split-K GEMMs of these 8-row problems add their slices with float atomics, and 12 guided steps amplify the round-off
    diffs = [float((a - e).abs().max()) for a, e in ((a1, e1), (a2, e2), (a3, e3))]
    assert max(diffs) < 5e-4, diffs
    assert float((a1 - a3).abs().max()) > 1e-3 and float((a1 - a2).abs().max()) > 1e-3
