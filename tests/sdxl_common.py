"""shared by the GPU and the emulator tests of the SDXL sampling loop: a small stand-in pipeline and the oracle run that mirrors one product call"""
import numpy as np
import torch

from oracle import sdxl_pipeline as osp


def small_pipe(sched, device, latent=16, dtype=torch.float16):
    from eeg_image_decode_amd.sdxl import DDIMScheduler, EulerAncestralDiscreteScheduler, SDXLShapedUNet, StandInSDXLPipeline
    unet = SDXLShapedUNet(stage_layers=(1, 1, 1, 1, 1), dtype=dtype, seed=5)
    pipe = StandInSDXLPipeline(unet, DDIMScheduler() if sched == "ddim" else EulerAncestralDiscreteScheduler(), device=device, dtype=dtype,
                               default_sample_size=latent)
    W = {k: v.detach().float().cpu().numpy().astype(np.float64) for k, v in unet.state_dict().items()}
    return pipe, W, ((1, 1, 1, 1, 1), 1.0)


def fp16_round(x):
    return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


def oracle_loop(pipe, W, cfg, sched, steps, guidance, emb, seed, low_level_latent=None, strength=1.0):
    """the oracle's sampling loop with the product's inputs: the empty-prompt embeddings of `pipe`, the same CPU generator stream (start latents first,
    then one ancestral-noise tensor per step), fp16 rounding at the product's storage points"""
    B = emb.shape[0]
    L = pipe.default_sample_size
    gen = torch.Generator().manual_seed(seed)
    s = osp.DDIM(steps) if sched == "ddim" else osp.EulerAncestral(steps)
    t_start = 0
    if low_level_latent is None:
        lat0 = torch.randn(B, 4, L, L, generator=gen, dtype=torch.float16).numpy().astype(np.float64)
    else:
        t_start = max(steps - min(int(steps * strength), steps), 0)
        noise = torch.randn(low_level_latent.shape, generator=gen, dtype=torch.float16).numpy().astype(np.float64)
        lat0 = fp16_round(fp16_round(fp16_round(low_level_latent) * pipe.vae_scaling_factor) + noise)
    noises = [torch.randn(B, 4, L, L, generator=gen, dtype=torch.float16).numpy().astype(np.float64) for _ in range(steps)] if sched == "euler" else []
    text = np.repeat(pipe.empty_prompt_embeds.float().cpu().numpy().astype(np.float64), B, 0)
    pooled = np.repeat(pipe.empty_pooled_embeds.float().cpu().numpy().astype(np.float64), B, 0)
    side = float(L * 8)
    time_ids = np.array([side, side, 0.0, 0.0, side, side])
    return osp.sample_loop(W, cfg, s, lat0, text, pooled, time_ids, fp16_round(emb), guidance, noises, fp16_round, t_start=t_start)


class FakeAttention(torch.nn.Module):
    """the attributes of diffusers.models.attention_processor.Attention that a processor touches"""

    def __init__(self, dim, cross_dim, heads):
        super().__init__()
        nn = torch.nn
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(cross_dim, dim, bias=False)
        self.to_v = nn.Linear(cross_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0


def check_processor_kv_cache_never_serves_another_tensors_projections(device, dtype=torch.float16, dim=128, heads=2, cross=64, HW=64):
    """the K / V cache of HIPIPAdapterAttnProcessor must not be keyed on an ADDRESS.  diffusers hands the processor a fresh image-token
    tensor on every UNet forward; between two Generator4Embeds.generate() calls the allocator reuses the freed address for the next image's tokens
    (same shape, version 0) and an address key would serve the previous image's K_ip / V_ip.  Here: tokens are freed and new ones of the same shape
    written INTO THE SAME STORAGE ADDRESS (the worst case, forced) -- the output must follow the new tokens; inside begin/end_sampling_run the
    projections are computed once although every step passes a fresh tensor object, and nothing survives the run."""
    from eeg_image_decode_amd import sdxl
    torch.manual_seed(1)
    B = 2
    attn = FakeAttention(dim, cross, heads).to(device).to(dtype)
    proc = sdxl.HIPIPAdapterAttnProcessor(dim, cross, scale=1.0).to(device).to(dtype)
    hs = torch.randn(B, HW, dim, device=device, dtype=dtype)
    text = torch.randn(B, 77, cross, device=device, dtype=dtype)
    store = torch.empty(B, 1, 4, cross, device=device, dtype=dtype)           # one storage, reused: every `ip_*` below lives at the same address
    ip_a_vals, ip_b_vals = torch.randn_like(store), torch.randn_like(store)
    calls = []
    real = sdxl.linear16
    sdxl.linear16 = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            def run(vals):
                buf = store.detach()                                          # a NEW tensor object at the old address
                assert buf.data_ptr() == store.data_ptr() and buf._version == 0
                buf.untyped_storage().copy_(vals.untyped_storage())                                                   # (raw copy: version stays 0)
                assert buf._version == 0
                return proc(attn, hs, encoder_hidden_states=(text, [buf]))
            out_a = run(ip_a_vals).clone()
            out_b = run(ip_b_vals).clone()                                    # same address, shape, dtype, version -- different image
            fresh = sdxl.HIPIPAdapterAttnProcessor(dim, cross, scale=1.0).to(device).to(dtype)
            fresh.load_state_dict(proc.state_dict())
            want_b = fresh(attn, hs, encoder_hidden_states=(text, [ip_b_vals.clone()]))
            assert torch.equal(out_b, want_b), "stale K_ip / V_ip served for a new token tensor"
            assert not torch.equal(out_a, out_b)
            # same OBJECT again: cache hit (q + out projections only)
            keep = ip_b_vals.clone()
            proc(attn, hs, encoder_hidden_states=(text, [keep]))
            n0 = len(calls)
            proc(attn, hs, encoder_hidden_states=(text, [keep]))
            assert len(calls) - n0 == 2
            # run scope: fresh objects every step, projected once; the next run starts empty
            proc.begin_sampling_run()
            n0 = len(calls)
            o1 = proc(attn, hs, encoder_hidden_states=(text.clone(), [ip_a_vals.clone()]))
            o2 = proc(attn, hs, encoder_hidden_states=(text.clone(), [ip_a_vals.clone()]))
            assert len(calls) - n0 == 6 + 2 and torch.equal(o1, o2)
            proc.end_sampling_run()
            proc.begin_sampling_run()
            o3 = proc(attn, hs, encoder_hidden_states=(text.clone(), [ip_b_vals.clone()]))
            proc.end_sampling_run()
            assert torch.equal(o3, want_b) and not proc._kv_cache
    finally:
        sdxl.linear16 = real
