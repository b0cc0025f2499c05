"""Oracle: the SDXL sampling loop around the IP-Adapter cross-attention (numpy float64, optional 16-bit rounding at layer boundaries).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

What is restated, and from where:
  * the loop -- order of operations of Generation/custom_pipeline.py:244-385 (timesteps, latents * init_noise_sigma, [negative | positive]
    batch under classifier-free guidance with a ZERO negative image embedding :321-324, scale_model_input, unet, guidance mix :376-378,
    scheduler.step :384) and the img2img start of Generation/custom_pipeline_low_level.py:331-389 (timesteps[t_start:], latent * scaling + noise).
  * the two schedulers -- diffusers==0.30.0 (requirements.txt:8), NOT vendored by the reference and not installed here: PARITY UNPINNED.
    DDIMScheduler with SDXL-base's scheduler_config (scaled-linear betas 0.00085..0.012, leading spacing, steps_offset 1, eta 0, no clipping,
    set_alpha_to_one False); EulerAncestralDiscreteScheduler as sdxl-turbo configures it (trailing spacing).
  * the UNet -- there is NO reference restatement: the product's `SDXLShapedUNet` is a stand-in (SDXL-base's cross-attention shapes, random
    weights, no self-attention / ResNets); this file restates THAT forward so the HIP path can be checked end to end.
"""
import math

import numpy as np

from . import sdxl_attn


def alphas_cumprod(T=1000, b0=0.00085, b1=0.012):
    betas = np.linspace(b0 ** 0.5, b1 ** 0.5, T, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


class DDIM:
    init_noise_sigma = 1.0

    def __init__(self, n):
        self.acp = alphas_cumprod()
        self.ratio = 1000 // n
        self.timesteps = (np.arange(n) * self.ratio).round()[::-1].astype(np.int64) + 1
        self.i = 0

    def scale(self, x):
        return x

    def step(self, eps, x, noise=None):
        t = int(self.timesteps[self.i])
        tp = t - self.ratio
        a_t, a_p = self.acp[t], (self.acp[tp] if tp >= 0 else self.acp[0])
        x0 = (x - math.sqrt(1 - a_t) * eps) / math.sqrt(a_t)
        self.i += 1
        return math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * eps


class EulerAncestral:
    def __init__(self, n):
        acp = alphas_cumprod()
        ts = np.round(np.arange(1000, 0, -1000 / n)) - 1
        sig = ((1 - acp) / acp) ** 0.5
        self.sigmas = np.concatenate([sig[ts.astype(np.int64)], [0.0]])
        self.timesteps = ts.astype(np.int64)
        self.init_noise_sigma = float(math.sqrt(self.sigmas.max() ** 2 + 1))
        self.i = 0

    def scale(self, x):
        return x / math.sqrt(self.sigmas[self.i] ** 2 + 1)

    def step(self, eps, x, noise):
        s, sn = self.sigmas[self.i], self.sigmas[self.i + 1]
        x0 = x - s * eps
        s_up = math.sqrt(max(sn ** 2 * (s ** 2 - sn ** 2) / s ** 2, 0.0))
        s_down = math.sqrt(max(sn ** 2 - s_up ** 2, 0.0))
        d = (x - x0) / s
        self.i += 1
        return x + d * (s_down - s) + noise * s_up


def sinusoid(t, dim):
    half = dim // 2
    f = np.exp(-math.log(10000.0) * np.arange(half, dtype=np.float32) / half).astype(np.float32)
    arg = np.asarray(t, np.float32)[..., None] * f
    return np.concatenate([np.cos(arg), np.sin(arg)], -1)


def silu(x):
    return x / (1 + np.exp(-x))


def standin_unet(W, cfg, sample, t, text, added, rnd):
    """SDXLShapedUNet.forward.  W: dict name -> float64 array (the module's state_dict), cfg: (stage_layers, ip_scale), rnd: rounding applied where
    the product stores a 16-bit tensor."""
    stage_layers, ip_scale = cfg
    dims = (640, 1280, 1280, 1280, 640)
    B, _, L, _ = sample.shape
    l1, l2 = L // 2, L // 4
    lin = lambda x, w, b=None: rnd(x @ W[w].T + (W[b] if b else 0.0))
    temb = rnd(sinusoid(np.full(B, t, np.float32), 320).astype(np.float64))
    e = lin(rnd(silu(lin(temb, "time_w1"))), "time_w2")
    aug = np.concatenate([added["text_embeds"], rnd(sinusoid(added["time_ids"].reshape(-1), 256).reshape(B, -1).astype(np.float64))], -1)
    e = rnd(e + lin(rnd(silu(lin(aug, "add_w1"))), "add_w2"))
    emb = rnd(silu(e))
    ip = None
    if added.get("image_embeds") is not None:
        x = lin(added["image_embeds"], "image_proj", "image_proj_bias").reshape(B, -1, 2048)
        mu, var = x.mean(-1, keepdims=True), x.var(-1, keepdims=True)
        ip = rnd((x - mu) / np.sqrt(var + 1e-5) * W["image_ln_w"] + W["image_ln_b"])
    x = sample.reshape(B, 4, l1, 2, l1, 2).transpose(0, 2, 4, 1, 3, 5).reshape(B, l1 * l1, 16)
    h = rnd(x @ W["conv_in"][:, :16].T + lin(emb, "stage_t.0")[:, None, :])
    slot = [0]

    def run(n, h):
        for _ in range(n):
            p = f"slots.{slot[0]}."
            slot[0] += 1
            heads = h.shape[-1] // 64
            q = rnd(h @ W[p + "to_q"].T)
            k, v = rnd(text @ W[p + "to_k"].T), rnd(text @ W[p + "to_v"].T)
            kip = vip = None
            if ip is not None:
                kip, vip = rnd(ip @ W[p + "to_k_ip"].T), rnd(ip @ W[p + "to_v_ip"].T)
            a = rnd(sdxl_attn.cross_attention(q, k, v, heads, kip, vip, ip_scale))
            h = rnd(a @ W[p + "to_out"].T + W[p + "to_out_bias"] + h)
        return h

    h = run(stage_layers[0], h)
    h = h.reshape(B, l2, 2, l2, 2, 640).transpose(0, 1, 3, 2, 4, 5).reshape(B, l2 * l2, 2560)
    h = rnd(h @ W["down"].T + lin(emb, "stage_t.1")[:, None, :])
    for s in (1, 2, 3):
        h = run(stage_layers[s], h)
    h = rnd(h @ W["up"].T).reshape(B, l2, l2, 2, 2, 640).transpose(0, 1, 3, 2, 4, 5).reshape(B, l1 * l1, 640)
    h = rnd(h + lin(emb, "stage_t.2")[:, None, :])
    h = run(stage_layers[4], h)
    y = rnd(h @ W["conv_out"].T)[..., :16]
    return y.reshape(B, l1, l1, 4, 2, 2).transpose(0, 3, 1, 4, 2, 5).reshape(B, 4, L, L)


def sample_loop(W, cfg, sched, latents0, text, pooled, time_ids, image_embeds, guidance, noises, rnd, t_start=0):
    """generate_ip_adapter_embeds: latents0 = the start latents BEFORE the init_noise_sigma scaling (or the img2img start as is when t_start > 0 /
    latents0 already noised), noises = the per-step ancestral noise tensors in order.  Returns the final latents."""
    do_cfg = guidance > 1.0
    x = rnd(latents0 * (sched.init_noise_sigma if t_start == 0 else 1.0))
    sched.i = t_start
    B = x.shape[0]
    if do_cfg:
        text = np.concatenate([np.zeros_like(text), text])
        pooled = np.concatenate([np.zeros_like(pooled), pooled])
        if image_embeds is not None:
            image_embeds = np.concatenate([np.zeros_like(image_embeds), image_embeds])
    nb = 2 * B if do_cfg else B
    added = {"text_embeds": pooled, "time_ids": np.repeat(time_ids[None], nb, 0), "image_embeds": image_embeds}
    ni = 0
    for t in sched.timesteps[t_start:]:
        xin = np.concatenate([x, x]) if do_cfg else x
        xin = rnd(sched.scale(xin))
        eps = standin_unet(W, cfg, xin, int(t), text, added, rnd)
        if do_cfg:
            eu, ec = eps[:B], eps[B:]
            eps = eu + guidance * (ec - eu)
        nz = None
        if isinstance(sched, EulerAncestral):
            nz, ni = noises[ni], ni + 1
        x = rnd(sched.step(eps, x, nz))
    return x
