#!/usr/bin/env python3
"""Golden fixtures for the diffusion prior (rows E1-E3), produced by running the reference's
Generation/diffusion_prior.py in place.  `diffusers` is not installed: its pieces the reference imports are injected
as stubs built from oracle/prior.py's restatement (so the fixtures pin the REFERENCE'S use of them -- model
composition, training objective, CFG loop, scheduler call order -- not diffusers' own arithmetic, which stays
"parity unpinned")."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
SEED = 20260926

from eeg_image_decode_amd import synthetic as syn   # noqa: E402
from oracle import prior as oprior                   # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        assert flip_sin_to_cos is True and downscale_freq_shift == 0
        self.num_channels = num_channels

    def forward(self, t):
        return oprior.timestep_embedding(t, self.num_channels)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


def import_reference_prior():
    d = _stub("diffusers")
    d.models = _stub("diffusers.models")
    d.models.embeddings = _stub("diffusers.models.embeddings", Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding)
    d.schedulers = _stub("diffusers.schedulers", DDPMScheduler=oprior.DDPMSchedulerOracle)

    def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps):
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: oprior.cosine_with_warmup_lr(s, 1.0, num_warmup_steps, num_training_steps))
    d.optimization = _stub("diffusers.optimization", get_cosine_schedule_with_warmup=get_cosine_schedule_with_warmup)
    _stub("diffusers.pipelines")
    _stub("diffusers.pipelines.stable_diffusion_xl")
    _stub("diffusers.pipelines.stable_diffusion_xl.pipeline_stable_diffusion_xl", retrieve_timesteps=oprior.retrieve_timesteps)
    spec = importlib.util.spec_from_file_location("ref_prior", os.path.join(REF, "Generation", "diffusion_prior.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def gen_prior():
    ref = import_reference_prior()
    torch.manual_seed(0)
    m = ref.DiffusionPriorUNet(cond_dim=1024, dropout=0.1)
    sd0 = m.state_dict()
    spec = oprior.prior_state_spec()
    assert [k for k, _, _ in spec] == list(sd0.keys()), "oracle.prior_state_spec drifted from the reference state_dict"
    assert {k: list(s) for k, s, _ in spec} == {k: list(v.shape) for k, v in sd0.items()}
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 9675648
    with open(os.path.join(HERE, "prior_keys.json"), "w") as f:
        json.dump({"keys": {k: list(v.shape) for k, v in sd0.items()}, "n_params": n_params}, f, indent=0)
    state = syn.make_state(SEED + 20, spec)
    m.load_state_dict({k: t(v) for k, v in state.items()})
    out = {}
    N = 6
    x = t(syn.unit_features(SEED + 21, N, tag="px") * 8.0)
    c = t(syn.unit_features(SEED + 21, N, tag="pc") * 32.0)
    tt = torch.tensor([0, 5, 333, 999, 20, 980])
    m.eval()
    with torch.no_grad():
        out["eps_cond"] = m(x, tt, c).numpy()
        out["eps_uncond"] = m(x, tt).numpy()
        out["eps_float_t"] = m(x, tt.float(), c).numpy()
    # training objective with injected noise / timesteps, dropout off
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    m.train()
    sched = oprior.DDPMSchedulerOracle()
    Bn = 32
    h = t(syn.unit_features(SEED + 22, Bn, tag="ph") * 6.0)
    cc = t(syn.unit_features(SEED + 22, Bn, tag="pcc") * 32.0)
    noise = t(syn.eeg_batch(SEED + 22, Bn, 1, 1024)[:, 0])
    ts = torch.from_numpy(np.random.default_rng(SEED + 22).integers(0, 1000, Bn))
    pert = sched.add_noise(h, noise, ts)
    pred = m(pert, ts, cc)
    loss = ((pred - noise) ** 2).mean()
    loss.backward()
    out["train_loss"] = np.float32(loss.item())
    out["train_pred_head"] = pred.detach().numpy()[:, :64].copy()
    for k, p in m.named_parameters():
        out["gnorm:" + k] = np.float32(p.grad.norm().item())
    out["total_gnorm"] = np.float32(torch.sqrt(sum((p.grad ** 2).sum() for p in m.parameters())).item())
    # Pipe.train: 3 steps on a fixed 2-batch loader with torch RNG pinned; records the lr sequence and the weights drift
    m2 = ref.DiffusionPriorUNet(cond_dim=1024, dropout=0.0)
    m2.load_state_dict({k: t(v) for k, v in state.items()})
    pipe = ref.Pipe(m2, scheduler=oprior.DDPMSchedulerOracle(), device="cpu")
    data = [{"c_embedding": cc[:16], "h_embedding": h[:16]}, {"c_embedding": cc[16:], "h_embedding": h[16:]}]

    class Loader(list):
        pass
    torch.manual_seed(1234)
    before = {k: p.detach().clone() for k, p in m2.named_parameters()}
    pipe.train(Loader(data), num_epochs=2, learning_rate=1e-3)
    out["pipe_train_dnorm_total"] = np.float32(torch.sqrt(sum(((p.detach() - before[k]) ** 2).sum() for k, p in m2.named_parameters())).item())
    out["pipe_train_out_w_head"] = m2.output_layer.weight.detach().numpy()[:4, :16].copy()
    # Pipe.generate, batch 1, 50 steps, CFG 5.0, seeded CPU generator
    m.eval()
    pipe_g = ref.Pipe(m, scheduler=oprior.DDPMSchedulerOracle(), device="cpu")
    g = torch.Generator().manual_seed(77)
    hgen = pipe_g.generate(c_embeds=cc[:1], num_inference_steps=50, guidance_scale=5.0, generator=g)
    out["gen_final"] = hgen.detach().numpy()
    g = torch.Generator().manual_seed(78)
    out["gen_final_uncond_10steps"] = pipe_g.generate(c_embeds=None, num_inference_steps=10, guidance_scale=5.0, generator=g).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "prior.npz"), **out)


if __name__ == "__main__":
    gen_prior()
    for f in ("prior.npz", "prior_keys.json"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
