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
