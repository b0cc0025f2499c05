"""SDXL sampling side of the path (Generation/custom_pipeline.py, custom_pipeline_low_level.py) on HIP kernels.

What the reference does there: `generate_ip_adapter_embeds` is diffusers' SDXL sampling loop with
one extra input -- the EEG-predicted CLIP image embedding, fed to the UNet's cross-attention through the IP-Adapter -- and
`Generator4Embeds` wraps sdxl-turbo around it.  diffusers and every checkpoint are third-party and absent offline, so this module is
self-contained:

* cross_attention(...) / HIPIPAdapterAttnProcessor   the UNet cross-attention with the IP-Adapter branch in ONE kernel (csrc/cross_attn.hip);
                                                     q / k / v / out projections on the 16-bit MFMA GEMM (csrc/gemm16.hip), K and V of the
                                                     77 text + 4 image tokens projected ONCE per sampling run (they do not change between
                                                     denoising steps).  Drop-in for diffusers' IPAdapterAttnProcessor2_0 when diffusers exists.
* EulerAncestralDiscreteScheduler / DDIMScheduler    the two schedulers of the path, restated from diffusers 0.30.0 (parity unpinned); the
                                                     per-step update of the latents (+ classifier-free-guidance mix) is one kernel launch.
* generate_ip_adapter_embeds(pipe, ...)              the sampling loop with the reference's signature and order of operations
                                                     (custom_pipeline.py:244-385), plus the img2img start of custom_pipeline_low_level.py:331-389.
* SDXLShapedUNet                                     a randomly initialised STAND-IN for the UNet with SDXL-base's cross-attention stack:
                                                     70 transformer positions at the real shapes (10 x (HW/4, 640 ch, 10 heads), 60 x (HW/16,
                                                     1280 ch, 20 heads)), time / added-condition embedding, 2 x 2 patch merges between the
                                                     resolutions.  It has NO self-attention, ResNet blocks or trained weights: it exists so that
                                                     the loop, the schedulers and the attention kernels run and are measured end to end.
* StandInSDXLPipeline / Generator4Embeds             the wrapper (custom_pipeline.py:456-492).  With diffusers + checkpoints it drives the real
                                                     pipeline; offline it drives the stand-in and returns LATENTS (there is no VAE to decode).
"""
import math

import torch
import torch.nn as nn

from . import _abi
from ._lib import EegclipError, check, lib, raw_stream, require_cuda


def _stream():
    return raw_stream()


def _dt(t):
    if t.dtype == torch.float16:
        return _abi.DT_F16
    if t.dtype == torch.bfloat16:
        return _abi.DT_BF16
    raise EegclipError("the SDXL path runs in fp16 or bf16 (the pipeline dtype of the reference: custom_pipeline.py:459, custom_pipeline_low_level.py:576)")


def linear16(x, weight, bias=None, residual=None, r_div=0):
    """y = x @ weight.T (+ bias) (+ residual) on the 16-bit matrix cores.  x (..., K), weight (N, K) (nn.Linear layout), residual shaped like y,
    or (rows / r_div, N) with r_div > 0 (one row per block of r_div consecutive rows: a per-sample embedding).  N % 128 == 0, K % 64 == 0."""
    require_cuda(x, "x")
    dt = _dt(x)
    K = x.shape[-1]
    N = weight.shape[0]
    if weight.shape[1] != K or weight.dtype != x.dtype:
        raise EegclipError(f"linear16: weight {tuple(weight.shape)} {weight.dtype} does not match input (..., {K}) {x.dtype}")
    if N % 128 or K % 64:
        raise EegclipError(f"linear16 takes N % 128 == 0 and K % 64 == 0 (got N = {N}, K = {K}); pad the layer")
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 8:
        x2 = x2.contiguous()
    w = weight if weight.is_contiguous() else weight.contiguous()
    M = x2.shape[0]
    out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N)
        if r2.stride(1) != 1 or r2.stride(0) % 4:
            r2 = r2.contiguous()
        if r2.shape[0] != (M if r_div == 0 else (M + r_div - 1) // r_div):
            raise EegclipError("linear16: residual rows do not match")
    b = bias.contiguous() if bias is not None else None
    check(lib().eegclip_gemm16(x2.data_ptr(), x2.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), N, b.data_ptr() if b is not None else None,
                               r2.data_ptr() if r2 is not None else None, r2.stride(0) if r2 is not None else 0, int(r_div), M, N, K, dt, _stream()), "gemm16")
    return out.reshape(*x.shape[:-1], N)


def cross_attention(q, k, v, heads, k_ip=None, v_ip=None, ip_scale=1.0):
    """softmax(q k^T/8) v + ip_scale * softmax(q k_ip^T/8) v_ip, head_dim 64.  q (B,HW,C); k,v (B,S,C); k_ip,v_ip (B,S_ip,C)."""
    require_cuda(q, "q")
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise EegclipError("cross_attention runs in fp16 or bf16 (the SDXL pipeline dtype)")
    B, HW, C = q.shape
    if C != heads * 64:
        raise EegclipError(f"head_dim must be 64 (C={C}, heads={heads})")
    q, k, v = q.contiguous(), k.to(q.dtype).contiguous(), v.to(q.dtype).contiguous()
    S = k.shape[1]
    S_ip = 0
    kp = vp = None
    if k_ip is not None:
        k_ip, v_ip = k_ip.to(q.dtype).contiguous(), v_ip.to(q.dtype).contiguous()
        S_ip = k_ip.shape[1]
        kp, vp = k_ip.data_ptr(), v_ip.data_ptr()
    out = torch.empty_like(q)
    check(lib().eegclip_cross_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), kp, vp, out.data_ptr(), B, HW, heads, 64, S, S_ip,
                                       float(ip_scale), _abi.DT_F16 if q.dtype == torch.float16 else _abi.DT_BF16, _stream()), "cross_attn_fwd")
    return out


def _hip_linear_ok(lin, x):
    return lin.weight.shape[0] % 128 == 0 and lin.weight.shape[1] % 64 == 0 and lin.weight.dtype == x.dtype      # (linear16 itself rejects CPU tensors)


class HIPIPAdapterAttnProcessor(nn.Module):
    """Cross-attention processor with an optional IP-Adapter branch (to_k_ip / to_v_ip: Linear(cross_attention_dim -> hidden_size),
    the non-"plus" adapter with 4 image tokens; scale 1 as in the reference, custom_pipeline.py:476).  Every GEMM of the layer -- attn.to_q,
    to_k, to_v, to_out[0], to_k_ip, to_v_ip -- runs on csrc/gemm16.hip (no library GEMM); the K / V projections of the text and image tokens are
    cached: inside a sampling run (begin_sampling_run / end_sampling_run, driven by generate_ip_adapter_embeds) they are computed on the first
    denoising step only; outside one, only for the very same tensor object."""

    def __init__(self, hidden_size, cross_attention_dim=2048, num_tokens=4, scale=1.0, with_ip=True):
        super().__init__()
        self.hidden_size, self.cross_attention_dim, self.num_tokens, self.scale = hidden_size, cross_attention_dim, num_tokens, scale
        self.to_k_ip = nn.Linear(cross_attention_dim, hidden_size, bias=False) if with_ip else None
        self.to_v_ip = nn.Linear(cross_attention_dim, hidden_size, bias=False) if with_ip else None
        self._kv_cache = {}

    @staticmethod
    def _lin(layer, x, residual=None):
        if not _hip_linear_ok(layer, x):
            raise EegclipError(f"projection {tuple(layer.weight.shape)} {layer.weight.dtype} is outside the 16-bit GEMM's shapes (N % 128, K % 64, dtype of "
                               "the activations); this processor issues no library GEMM")
        return linear16(x, layer.weight, layer.bias, residual)

    def begin_sampling_run(self):
        """Called by the sampling loop before its first UNet forward: inside one run the text / image tokens are constants (custom_pipeline.py:296-373
        builds them before the loop), so K / V are projected on the first step and reused on the others even when the pipeline hands over a FRESH
        token tensor every step (diffusers' encoder_hid_proj does).  Nothing cached survives the call: a new run never sees an old run's K / V."""
        self._kv_cache.clear()
        self._run_scope = True

    def end_sampling_run(self):
        self._kv_cache.clear()
        self._run_scope = False

    def _projected(self, tag, lin_k, lin_v, src, make_tokens):
        """K, V of the token tensor `src` (make_tokens() -> the (B, S, cross_dim) matrix in the activation dtype).  Outside a sampling run they are
        reused only for THE SAME tensor object at the same version (the entry holds a reference to it, so its address cannot be recycled for
        another tensor while the entry lives -- keying on data_ptr() could hand the K / V of a freed tensor to the next one allocated at its
        address); inside a run (begin_sampling_run) the first projection of the run is reused for any OTHER tensor of that shape (pipelines rebuild
        the constant token tensor every step), but the SAME object must still be at the version it was projected at: a callback that edits
        prompt_embeds / image_embeds in place between steps gets fresh K / V instead of silently stale ones."""
        hit = self._kv_cache.get(tag)
        wkey = (lin_k.weight.data_ptr(), lin_k.weight._version, lin_v.weight.data_ptr(), lin_v.weight._version)
        if hit is not None and hit[2] == wkey and hit[0].shape == src.shape and hit[0].dtype == src.dtype:
            same_obj = hit[0] is src
            if (same_obj and hit[1] == src._version) or (not same_obj and getattr(self, "_run_scope", False)):
                return hit[3], hit[4]
        tokens = make_tokens()
        k, v = self._lin(lin_k, tokens), self._lin(lin_v, tokens)
        self._kv_cache[tag] = (src, src._version, wkey, k, v)
        return k, v

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0, ip_adapter_masks=None, **kw):
        if encoder_hidden_states is None:
            raise EegclipError("HIPIPAdapterAttnProcessor is for CROSS-attention layers (attn2); keep the stock processor on self-attention")
        if attention_mask is not None or ip_adapter_masks is not None:
            raise EegclipError("attention masks are not used on this path (empty prompt, single image embedding)")
        residual = hidden_states
        shape4 = hidden_states.shape if hidden_states.dim() == 4 else None
        if shape4 is not None:
            b, c, hh, ww = shape4
            hidden_states = hidden_states.view(b, c, hh * ww).transpose(1, 2)
        ip_tokens = None
        if isinstance(encoder_hidden_states, (tuple, list)):                 # diffusers >= 0.25: (text states, [image token tensors])
            encoder_hidden_states, ip_list = encoder_hidden_states
            ip_tokens = ip_list[0] if isinstance(ip_list, (list, tuple)) else ip_list
        elif self.to_k_ip is not None and encoder_hidden_states.shape[1] > 77:   # older layout: image tokens concatenated after the 77 text tokens
            end = encoder_hidden_states.shape[1] - self.num_tokens
            encoder_hidden_states, ip_tokens = encoder_hidden_states[:, :end], encoder_hidden_states[:, end:]
        text_src = encoder_hidden_states                                      # (identity of what the pipeline handed over: the cache key)
        q = self._lin(attn.to_q, hidden_states)

        def text_tokens():
            e = encoder_hidden_states
            if getattr(attn, "norm_cross", None):
                e = attn.norm_encoder_hidden_states(e)
            return e.to(q.dtype)
        k, v = self._projected("text", attn.to_k, attn.to_v, text_src, text_tokens)
        k_ip = v_ip = None
        if ip_tokens is not None and self.to_k_ip is not None:
            ip_src = ip_tokens
            k_ip, v_ip = self._projected("ip", self.to_k_ip, self.to_v_ip, ip_src,
                                         lambda: (ip_src.flatten(1, 2) if ip_src.dim() == 4 else ip_src).to(q.dtype))      # (B, n_images, tokens, dim)
        out = cross_attention(q, k, v, attn.heads, k_ip, v_ip, self.scale)
        fuse_res = bool(getattr(attn, "residual_connection", False)) and shape4 is None and getattr(attn, "rescale_output_factor", 1.0) == 1.0
        out = self._lin(attn.to_out[0], out, residual if fuse_res else None)
        out = attn.to_out[1](out)
        if shape4 is not None:
            out = out.transpose(-1, -2).reshape(shape4)
        if getattr(attn, "residual_connection", False) and not fuse_res:
            out = out + residual
        return out / getattr(attn, "rescale_output_factor", 1.0)


def install_cross_attention_processors(unet, scale=1.0):
    """Swap every cross-attention (attn2) processor of a diffusers UNet2DConditionModel for the HIP one, carrying over the
    IP-Adapter to_k_ip / to_v_ip weights if they are already loaded."""
    procs = {}
    for name, old in unet.attn_processors.items():
        if ".attn2." not in name:
            procs[name] = old
            continue
        hidden = unet.get_submodule(name.rsplit(".processor", 1)[0]).to_q.out_features
        new = HIPIPAdapterAttnProcessor(hidden, unet.config.cross_attention_dim, scale=scale, with_ip=hasattr(old, "to_k_ip"))
        if hasattr(old, "to_k_ip"):
            new.to_k_ip.weight.data = old.to_k_ip[0].weight.data
            new.to_v_ip.weight.data = old.to_v_ip[0].weight.data
        procs[name] = new.to(unet.device, unet.dtype)
    unet.set_attn_processor(procs)
    return unet


# ---------------------------------------------------------------------------------------------------------------------- schedulers
def _scaled_linear_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2          # SDXL: "scaled_linear"
    return torch.cumprod(1.0 - betas, dim=0)


class _SchedulerBase:
    order = 1

    class _Cfg:
        num_train_timesteps = 1000

    def __init__(self):
        self.config = self._Cfg()
        self.alphas_cumprod = _scaled_linear_alphas_cumprod()
        self.timesteps = None
        self.num_inference_steps = None

    def _launch(self, x, eps_u, eps_c, noise, guidance, cx, ce, cn, in_scale, want_scaled):
        require_cuda(x, "latents")
        x = x.contiguous()
        out = torch.empty_like(x)
        scaled = torch.empty_like(x) if want_scaled else None
        check(lib().eegclip_sampler_step(x.data_ptr(), eps_u.contiguous().data_ptr(), eps_c.contiguous().data_ptr() if eps_c is not None else None,
                                         noise.contiguous().data_ptr() if noise is not None else None, out.data_ptr(),
                                         scaled.data_ptr() if scaled is not None else None, float(guidance), float(cx), float(ce), float(cn), float(in_scale),
                                         x.numel(), _dt(x), _stream()), "sampler_step")
        return out, scaled


class DDIMScheduler(_SchedulerBase):
    """diffusers 0.30.0 DDIMScheduler with the SDXL-base scheduler_config values, restated (parity unpinned): scaled-linear betas 0.00085..0.012,
    epsilon prediction, eta = 0, clip_sample False, set_alpha_to_one False, timestep_spacing "leading", steps_offset 1."""
    init_noise_sigma = 1.0

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        self.timesteps = ((torch.arange(0, num_inference_steps) * ratio).round().flip(0) + 1).long()      # kept on the host: no sync per step
        self._ratio = ratio

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coefficients(self, t):
        """x_{t-1} = cx * x_t + ce * eps (eta = 0).  With x0 = (x - sqrt(1 - a_t) eps) / sqrt(a_t):  x' = sqrt(a_p) x0 + sqrt(1 - a_p) eps."""
        t = int(t)
        tp = t - self._ratio
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[tp]) if tp >= 0 else float(self.alphas_cumprod[0])        # set_alpha_to_one = False: final_alpha_cumprod = a[0]
        cx = math.sqrt(a_p / a_t)
        ce = math.sqrt(1.0 - a_p) - math.sqrt(a_p * (1.0 - a_t) / a_t)
        return cx, ce, 0.0

    def step(self, model_output, timestep, sample, generator=None, return_dict=False, model_output_uncond=None, guidance_scale=0.0, **kw):
        """model_output_uncond (+ guidance_scale): the classifier-free-guidance mix fused into the same kernel (model_output = conditional)"""
        cx, ce, cn = self.coefficients(timestep)
        if model_output_uncond is not None:
            out, _ = self._launch(sample, model_output_uncond, model_output, None, guidance_scale, cx, ce, cn, 1.0, False)
        else:
            out, _ = self._launch(sample, model_output, None, None, 0.0, cx, ce, cn, 1.0, False)
        return (out,)


class EulerAncestralDiscreteScheduler(_SchedulerBase):
    """diffusers 0.30.0 EulerAncestralDiscreteScheduler as sdxl-turbo configures it, restated (parity unpinned): scaled-linear betas, epsilon
    prediction, timestep_spacing "trailing", sigmas = sqrt((1 - a) / a) interpolated at the timesteps with a final 0."""

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        self.num_inference_steps = num_inference_steps
        T = self.config.num_train_timesteps
        ts = torch.round(torch.arange(T, 0, -T / num_inference_steps, dtype=torch.float64)) - 1          # trailing: 999, ... for 1 step: [999]
        sig_all = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        sig = sig_all[ts.long()]
        self.sigmas = torch.cat([sig, torch.zeros(1, dtype=torch.float64)])
        self.timesteps = ts.long()
        self._step_index = 0

    @property
    def init_noise_sigma(self):
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)        # "trailing" spacing: sqrt(sigma_max^2 + 1)

    def scale_model_input(self, sample, timestep=None):
        s = float(self.sigmas[self._step_index])
        return sample * (1.0 / math.sqrt(s * s + 1.0))

    def coefficients(self, i):
        s, sn = float(self.sigmas[i]), float(self.sigmas[i + 1])
        s_up = math.sqrt(max(sn * sn * (s * s - sn * sn) / (s * s), 0.0))
        s_down = math.sqrt(max(sn * sn - s_up * s_up, 0.0))
        return 1.0, s_down - s, s_up          # x' = x + eps (sigma_down - sigma) + noise sigma_up     (derivative = eps for epsilon prediction)

    def step(self, model_output, timestep, sample, generator=None, return_dict=False, model_output_uncond=None, guidance_scale=0.0, **kw):
        i = self._step_index
        cx, ce, cn = self.coefficients(i)
        gdev = generator.device if generator is not None else sample.device
        noise = torch.randn(sample.shape, generator=generator, device=gdev, dtype=sample.dtype).to(sample.device)      # randn_tensor of the reference
        if model_output_uncond is not None:
            out, _ = self._launch(sample, model_output_uncond, model_output, noise, guidance_scale, cx, ce, cn, 1.0, False)
        else:
            out, _ = self._launch(sample, model_output, None, noise, 0.0, cx, ce, cn, 1.0, False)
        self._step_index += 1
        return (out,)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, **kw):
    scheduler.set_timesteps(num_inference_steps, device=device)
    return scheduler.timesteps, num_inference_steps


# ------------------------------------------------------------------------------------------------------------- stand-in UNet
def _sinusoid(t, dim, flip_sin_to_cos=True):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] of t * exp(-ln(1e4) i / (dim/2))"""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    arg = t.float()[..., None] * freqs
    return torch.cat([arg.cos(), arg.sin()] if flip_sin_to_cos else [arg.sin(), arg.cos()], dim=-1)


class _AttnSlot(nn.Module):
    """the parameters of one cross-attention position (diffusers Attention attribute names)"""

    def __init__(self, dim, cross_dim, n_layers_total, dtype):
        super().__init__()
        self.heads = dim // 64
        g = lambda *s, scale: nn.Parameter((torch.randn(*s) * scale).to(dtype), requires_grad=False)
        self.to_q = g(dim, dim, scale=dim ** -0.5)
        self.to_k = g(dim, cross_dim, scale=cross_dim ** -0.5)
        self.to_v = g(dim, cross_dim, scale=cross_dim ** -0.5)
        self.to_k_ip = g(dim, cross_dim, scale=cross_dim ** -0.5)
        self.to_v_ip = g(dim, cross_dim, scale=cross_dim ** -0.5)
        self.to_out = g(dim, dim, scale=dim ** -0.5 / math.sqrt(n_layers_total))      # small residual updates: activations stay in 16-bit range
        self.to_out_bias = nn.Parameter(torch.zeros(dim, dtype=dtype), requires_grad=False)


class SDXLShapedUNet(nn.Module):
    """Stand-in with SDXL-base's cross-attention stack (see the module docstring: NOT the SDXL UNet).  forward() has the call signature the
    pipeline uses (custom_pipeline.py:365-373).  Latents (B, 4, L, L): stage 1 works on (L/2)^2 tokens of 640 channels (10 heads), stage 2 on
    (L/4)^2 tokens of 1280 channels (20 heads); layers per stage default to SDXL-base's transformer counts (4 + 20 down, 10 mid, 30 + 6 up)."""

    def __init__(self, stage_layers=(4, 20, 10, 30, 6), cross_attention_dim=2048, ip_tokens=4, ip_scale=1.0, dtype=torch.float16, seed=0):
        super().__init__()

        class _C:
            in_channels = 4
            sample_size = 128
            time_cond_proj_dim = None
            addition_time_embed_dim = 256
        self.config = _C()
        self.config.cross_attention_dim = cross_attention_dim
        self.dtype_ = dtype
        self.ip_tokens, self.ip_scale = ip_tokens, ip_scale
        self.stage_layers = tuple(stage_layers)
        self.stage_dims = (640, 1280, 1280, 1280, 640)
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        n_total = sum(stage_layers)
        g = lambda *s, scale: nn.Parameter((torch.randn(*s) * scale).to(dtype), requires_grad=False)
        self.slots = nn.ModuleList([_AttnSlot(d, cross_attention_dim, n_total, dtype) for d, n in zip(self.stage_dims, stage_layers) for _ in range(n)])
        # time + added-condition embedding (diffusers: time_embedding(320 -> 1280), add_embedding(2816 -> 1280))
        self.time_w1, self.time_w2 = g(1280, 320, scale=320 ** -0.5), g(1280, 1280, scale=1280 ** -0.5)
        self.add_w1, self.add_w2 = g(1280, 2816, scale=2816 ** -0.5), g(1280, 1280, scale=1280 ** -0.5)
        self.stage_t = nn.ParameterList([g(d, 1280, scale=0.1 * 1280 ** -0.5) for d in (640, 1280, 640)])        # embedding -> channels, per resolution entry
        # 2 x 2 patch transitions: latent patches -> 640 | 4 x 640 -> 1280 | 1280 -> 4 x 640 | 640 -> latent patches (N padded to 128)
        self.conv_in = g(640, 64, scale=16 ** -0.5)                      # K = 16 real inputs (4 ch x 2 x 2), zero-padded to 64
        self.down = g(1280, 2560, scale=2560 ** -0.5)
        self.up = g(2560, 1280, scale=1280 ** -0.5)
        self.conv_out = g(128, 640, scale=640 ** -0.5)                   # N = 16 real outputs, padded to 128
        # IP-Adapter image projection (diffusers ImageProjection: Linear(1024 -> 4 * 2048) + LayerNorm(2048))
        self.image_proj = g(ip_tokens * cross_attention_dim, 1024, scale=1024 ** -0.5)
        self.image_proj_bias = nn.Parameter(torch.zeros(ip_tokens * cross_attention_dim, dtype=dtype), requires_grad=False)
        self.image_ln_w = nn.Parameter(torch.ones(cross_attention_dim, dtype=torch.float32), requires_grad=False)
        self.image_ln_b = nn.Parameter(torch.zeros(cross_attention_dim, dtype=torch.float32), requires_grad=False)
        torch.random.set_rng_state(gen_state)
        self._kv = None

    @property
    def dtype(self):
        return self.dtype_

    @property
    def device(self):
        return self.conv_in.device

    # ---- hoisted, once per sampling run: the keys / values of every layer (the text and image tokens do not change between steps)
    def image_tokens(self, image_embeds):
        """(B, 1024) -> (B, 4, 2048): Linear + LayerNorm.  (The LayerNorm of the 4 B token rows runs once per sampling run, outside the denoising
        loop: a torch op, fp32 arithmetic.)"""
        x = linear16(image_embeds.to(self.dtype_), self.image_proj, self.image_proj_bias)
        rows = x.reshape(-1, self.config.cross_attention_dim).float()
        y = torch.nn.functional.layer_norm(rows, (rows.shape[1],), self.image_ln_w, self.image_ln_b, 1e-5)
        return y.to(self.dtype_).reshape(image_embeds.shape[0], self.ip_tokens, -1)

    def precompute(self, encoder_hidden_states, image_embeds=None):
        text = encoder_hidden_states.to(self.dtype_).contiguous()
        ip = self.image_tokens(image_embeds) if image_embeds is not None else None
        B = text.shape[0]
        kv = []
        for s in self.slots:
            k, v = linear16(text, s.to_k), linear16(text, s.to_v)
            kip = vip = None
            if ip is not None:
                kip, vip = linear16(ip, s.to_k_ip), linear16(ip, s.to_v_ip)
            kv.append((k, v, kip, vip))
        # (the entry keeps the token tensors themselves: identity, not data_ptr(), decides a hit -- an address can be recycled by the allocator)
        self._kv = (encoder_hidden_states, encoder_hidden_states._version, image_embeds, None if image_embeds is None else image_embeds._version, B, kv)
        return self

    def _kv_for(self, encoder_hidden_states, image_embeds):
        c = self._kv
        if c is None or c[0] is not encoder_hidden_states or c[1] != encoder_hidden_states._version or c[2] is not image_embeds or \
                c[3] != (None if image_embeds is None else image_embeds._version) or c[4] != encoder_hidden_states.shape[0]:
            self.precompute(encoder_hidden_states, image_embeds)
        return self._kv[5]

    # ---- forward
    def embedding(self, timestep, B, added_cond_kwargs):
        dev = self.device
        t = torch.as_tensor(timestep, device=dev).reshape(-1).float().expand(B)
        temb = _sinusoid(t, 320).to(self.dtype_)
        e = linear16(torch.nn.functional.silu(linear16(temb, self.time_w1)), self.time_w2)
        text_embeds = added_cond_kwargs["text_embeds"].to(device=dev, dtype=self.dtype_)
        time_ids = added_cond_kwargs["time_ids"].to(device=dev)
        aug = torch.cat([text_embeds, _sinusoid(time_ids.flatten(), 256).reshape(B, -1).to(self.dtype_)], dim=-1)          # (B, 1280 + 6 * 256)
        e = e + linear16(torch.nn.functional.silu(linear16(aug, self.add_w1)), self.add_w2)
        return torch.nn.functional.silu(e)

    def forward(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                return_dict=False, **kw):
        require_cuda(sample, "sample")
        B, Cc, L, _ = sample.shape
        if Cc != 4 or L % 4:
            raise EegclipError("latents must be (B, 4, L, L) with L a multiple of 4")
        image_embeds = (added_cond_kwargs or {}).get("image_embeds")
        if isinstance(image_embeds, (list, tuple)):
            image_embeds = image_embeds[0]
        if image_embeds is not None and image_embeds.dim() == 3:
            image_embeds = image_embeds[:, 0]
        kv = self._kv_for(encoder_hidden_states, image_embeds)
        emb = self.embedding(timestep, B, added_cond_kwargs)                                # (B, 1280), SiLU applied
        l1, l2 = L // 2, L // 4
        # latents -> 2 x 2 patches -> (B, l1^2, 16 -> 64 zero padded)
        x = sample.to(self.dtype_).reshape(B, 4, l1, 2, l1, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, l1 * l1, 16)
        x = torch.nn.functional.pad(x, (0, 48))
        h = linear16(x, self.conv_in, None, linear16(emb, self.stage_t[0]), r_div=l1 * l1)
        it = iter(range(len(self.slots)))

        def run(n_layers, h):
            for _ in range(n_layers):
                i = next(it)
                s = self.slots[i]
                k, v, kip, vip = kv[i]
                a = cross_attention(linear16(h, s.to_q), k, v, s.heads, kip, vip, self.ip_scale)
                h = linear16(a, s.to_out, s.to_out_bias, h)                                  # residual fused into the output projection
            return h

        h = run(self.stage_layers[0], h)
        h = h.reshape(B, l2, 2, l2, 2, 640).permute(0, 1, 3, 2, 4, 5).reshape(B, l2 * l2, 2560)
        h = linear16(h, self.down, None, linear16(emb, self.stage_t[1]), r_div=l2 * l2)
        h = run(self.stage_layers[1], h)
        h = run(self.stage_layers[2], h)
        h = run(self.stage_layers[3], h)
        h = linear16(h, self.up)                                                              # (B, l2^2, 4 * 640)
        h = h.reshape(B, l2, l2, 2, 2, 640).permute(0, 1, 3, 2, 4, 5).reshape(B * l1 * l1, 640)
        h = (h.reshape(B, l1 * l1, 640) + linear16(emb, self.stage_t[2])[:, None, :]).contiguous()
        h = run(self.stage_layers[4], h)
        y = linear16(h, self.conv_out)[..., :16]
        out = y.reshape(B, l1, l1, 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(B, 4, L, L)
        return (out,)


# ------------------------------------------------------------------------------------------------------------- pipeline + loop
class _Output:
    def __init__(self, images):
        self.images = images


class StandInSDXLPipeline:
    """What `generate_ip_adapter_embeds` needs from a diffusers StableDiffusionXLPipeline, around the stand-in UNet: scheduler, empty-prompt
    embeddings (there is no text encoder offline: prompt '' maps to fixed embeddings, like the reference's constant empty prompt), micro-conditioning
    ids.  vae: a vae.SDXLShapedVAE (round 6: SDXL's VAE layout on csrc/vae.hip) -- `low_level_image` is then encoded by it
    (custom_pipeline_low_level.py:8-31) and output_type "pt" / "np" decodes the final latents (custom_pipeline.py:421 + image_processor.postprocess:
    image / 2 + 0.5 clamped to [0, 1]); without one, output_type must be "latent" unless `vae_decode` / `vae_encode` callables are supplied."""

    def __init__(self, unet=None, scheduler=None, device="cuda", dtype=torch.float16, default_sample_size=64, vae_decode=None, vae_encode=None, vae=None):
        self.unet = (unet if unet is not None else SDXLShapedUNet(dtype=dtype)).to(device)
        self.scheduler = scheduler if scheduler is not None else EulerAncestralDiscreteScheduler()
        self.device, self.dtype = device, dtype
        self.default_sample_size = default_sample_size           # sdxl-turbo: 512 px = 64 latent; SDXL-base: 128
        self.vae_scale_factor = 8
        self.vae_scaling_factor = 0.13025                        # SDXL VAE config.scaling_factor
        self.vae = vae.to(device) if vae is not None else None
        if self.vae is not None:
            self.vae_scaling_factor = self.vae.scaling_factor
            vae_decode = vae_decode or (lambda lat: self.vae.decode(lat))
            vae_encode = vae_encode or (lambda img, gen: self.vae.encode(img, generator=gen))
        self.vae_decode, self.vae_encode = vae_decode, vae_encode
        g = torch.Generator().manual_seed(1234)
        self.empty_prompt_embeds = (torch.randn(1, 77, 2048, generator=g) * 0.5).to(device=device, dtype=dtype)
        self.empty_pooled_embeds = (torch.randn(1, 1280, generator=g) * 0.5).to(device=device, dtype=dtype)
        self.text_encoder_projection_dim = 1280

    def encode_prompt(self, prompt, batch_size, prompt_embeds=None, pooled_prompt_embeds=None, negative_prompt_embeds=None,
                      negative_pooled_prompt_embeds=None, do_classifier_free_guidance=False):
        if prompt_embeds is None:
            if prompt not in (None, "", [""] * batch_size):
                raise EegclipError("no text encoder in the offline build: pass prompt_embeds / pooled_prompt_embeds, or the empty prompt the reference uses")
            prompt_embeds = self.empty_prompt_embeds.expand(batch_size, -1, -1)
            pooled_prompt_embeds = self.empty_pooled_embeds.expand(batch_size, -1)
        if do_classifier_free_guidance and negative_prompt_embeds is None:       # (an empty negative prompt embeds to zeros under force_zeros_for_empty_prompt)
            negative_prompt_embeds = torch.zeros_like(prompt_embeds)
            negative_pooled_prompt_embeds = torch.zeros_like(pooled_prompt_embeds)
        return prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds

    def prepare_latents(self, batch_size, channels, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device=device, dtype=dtype)
        return latents * self.scheduler.init_noise_sigma

    def prepare_latents_latent2img(self, latents, dtype, device, generator):
        """custom_pipeline_low_level.py:33-53: start from a low-level latent (scaled like a VAE encoding) plus unit noise"""
        latents = latents.to(device, dtype=dtype) * self.vae_scaling_factor
        gdev = generator.device if generator is not None else device
        noise = torch.randn(latents.shape, generator=generator, device=gdev, dtype=dtype).to(device)
        return latents + noise

    def prepare_latents_img2img(self, image, dtype, device, generator):
        """custom_pipeline_low_level.py:8-31: VAE-encode the low-level image, scale, add unit noise"""
        if self.vae_encode is None:
            raise EegclipError("low_level_image needs a VAE encoder (absent offline): pass vae_encode, or use low_level_latent")
        return self.prepare_latents_latent2img(self.vae_encode(image.to(device=device, dtype=dtype), generator), dtype, device, generator)

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, dtype):
        return torch.tensor([list(original_size + crops_coords_top_left + target_size)], dtype=dtype)

    def generate_ip_adapter_embeds(self, *a, **k):
        return generate_ip_adapter_embeds(self, *a, **k)


@torch.no_grad()
def generate_ip_adapter_embeds(self, prompt=None, prompt_2=None, height=None, width=None, num_inference_steps=50, timesteps=None, denoising_end=None,
                               guidance_scale=5.0, negative_prompt=None, negative_prompt_2=None, num_images_per_prompt=1, eta=0.0, generator=None,
                               latents=None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                               negative_pooled_prompt_embeds=None, ip_adapter_image=None, ip_adapter_embeds=None, output_type="latent", return_dict=True,
                               cross_attention_kwargs=None, guidance_rescale=0.0, original_size=None, crops_coords_top_left=(0, 0), target_size=None,
                               negative_original_size=None, negative_crops_coords_top_left=(0, 0), negative_target_size=None, clip_skip=None,
                               callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",), img2img_strength=1.0, low_level_image=None,
                               low_level_latent=None, **kwargs):
    """The reference's sampling loop (Generation/custom_pipeline.py:5-442; the three low-level arguments: custom_pipeline_low_level.py:56-560) over a
    pipeline object `self` (StandInSDXLPipeline here).  Same order of operations: encode prompt -> timesteps (-> img2img start) -> latents ->
    micro-conditioning ids -> [negative | positive] batch under classifier-free guidance, zeros as the negative image embedding -> loop:
    scale_model_input, unet, guidance mix, scheduler.step.  The guidance mix and the scheduler update are ONE kernel launch per step."""
    if ip_adapter_image is not None:
        raise EegclipError("ip_adapter_image needs the CLIP image encoder (absent offline); pass ip_adapter_embeds, as the reference's Generator4Embeds does")
    if guidance_rescale and guidance_rescale > 0.0:
        raise EegclipError("guidance_rescale is not used on this path (the reference passes 0)")
    height = height or self.default_sample_size * self.vae_scale_factor
    width = width or self.default_sample_size * self.vae_scale_factor
    original_size = original_size or (height, width)
    target_size = target_size or (height, width)
    do_cfg = guidance_scale > 1.0 and getattr(self.unet.config, "time_cond_proj_dim", None) is None          # diffusers: do_classifier_free_guidance
    if prompt is not None and isinstance(prompt, str):
        batch_size = 1
    elif prompt is not None and isinstance(prompt, list):
        batch_size = len(prompt)
    else:
        batch_size = prompt_embeds.shape[0]
    if ip_adapter_embeds is not None and prompt_embeds is None and ip_adapter_embeds.shape[0] != batch_size and isinstance(prompt, str):
        batch_size = ip_adapter_embeds.shape[0]                  # several image embeddings with the one (empty) prompt: one image each
    device, dtype = self.device, self.dtype
    prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = self.encode_prompt(
        prompt, batch_size, prompt_embeds, pooled_prompt_embeds, negative_prompt_embeds, negative_pooled_prompt_embeds, do_cfg)
    # 4. timesteps (+ the img2img start of the low-level variant: skip the first (1 - strength) fraction of the schedule)
    timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps)
    t_start = 0
    if low_level_image is not None or low_level_latent is not None:
        init_timestep = min(int(num_inference_steps * img2img_strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        timesteps = timesteps[t_start:]
        if hasattr(self.scheduler, "_step_index"):
            self.scheduler._step_index = t_start
    # 5. latents
    n_img = batch_size * num_images_per_prompt
    if low_level_image is not None and low_level_latent is None:
        latents = self.prepare_latents_img2img(low_level_image, dtype, device, generator)
    elif low_level_latent is not None:
        latents = self.prepare_latents_latent2img(low_level_latent, dtype, device, generator)
    else:
        latents = self.prepare_latents(n_img, self.unet.config.in_channels, height, width, dtype, device, generator, latents)
    # 7. added time ids & embeddings
    add_text_embeds = pooled_prompt_embeds
    add_time_ids = self._get_add_time_ids(original_size, tuple(crops_coords_top_left), target_size, dtype)
    if negative_original_size is not None and negative_target_size is not None:
        negative_add_time_ids = self._get_add_time_ids(negative_original_size, tuple(negative_crops_coords_top_left), negative_target_size, dtype)
    else:
        negative_add_time_ids = add_time_ids
    if do_cfg:
        prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
        add_text_embeds = torch.cat([negative_pooled_prompt_embeds, add_text_embeds], dim=0)
        add_time_ids = torch.cat([negative_add_time_ids, add_time_ids], dim=0)
    prompt_embeds = prompt_embeds.to(device).contiguous()
    add_text_embeds = add_text_embeds.to(device)
    add_time_ids = add_time_ids.to(device).repeat(n_img, 1)
    image_embeds = None
    if ip_adapter_embeds is not None:
        image_embeds = ip_adapter_embeds.to(device=device, dtype=prompt_embeds.dtype)
        if do_cfg:
            image_embeds = torch.cat([torch.zeros_like(image_embeds), image_embeds])
        image_embeds = image_embeds.contiguous()
    if denoising_end is not None and isinstance(denoising_end, float) and 0 < denoising_end < 1:
        cutoff = int(round(self.scheduler.config.num_train_timesteps - denoising_end * self.scheduler.config.num_train_timesteps))
        timesteps = timesteps[:len([ts for ts in timesteps.tolist() if ts >= cutoff])]
    if hasattr(self.unet, "precompute"):
        self.unet.precompute(prompt_embeds, image_embeds)        # K / V of all layers: once per run, not once per step
    run_scoped = [p for p in getattr(self.unet, "attn_processors", {}).values() if hasattr(p, "begin_sampling_run")]
    for p in run_scoped:                                         # (diffusers UNet with the HIP processors installed: per-run K / V caches)
        p.begin_sampling_run()
    added = {"text_embeds": add_text_embeds, "time_ids": add_time_ids}
    if image_embeds is not None:
        added["image_embeds"] = image_embeds
    # 8. denoising loop
    try:
        for i, t in enumerate(timesteps.tolist()):
            model_in = torch.cat([latents] * 2) if do_cfg else latents
            model_in = self.scheduler.scale_model_input(model_in, t)
            noise_pred = self.unet(model_in, t, encoder_hidden_states=prompt_embeds, timestep_cond=None, cross_attention_kwargs=cross_attention_kwargs,
                                   added_cond_kwargs=added, return_dict=False)[0]
            if do_cfg:
                eps_u, eps_c = noise_pred.chunk(2)
                latents = self.scheduler.step(eps_c, t, latents, generator=generator, model_output_uncond=eps_u, guidance_scale=guidance_scale)[0]
            else:
                latents = self.scheduler.step(noise_pred, t, latents, generator=generator)[0]
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": latents})
                latents = out.pop("latents", latents)
    finally:
        for p in run_scoped:
            p.end_sampling_run()
    if output_type == "latent":
        image = latents
    else:
        if self.vae_decode is None:
            raise EegclipError("decoding to an image needs the SDXL VAE (absent offline): use output_type='latent' or supply vae_decode")
        image = self.vae_decode(latents / self.vae_scaling_factor)
        if output_type in ("pt", "np", "pil"):                 # VaeImageProcessor.postprocess: denormalise to [0, 1]; "np" / "pil": (N, H, W, 3) on the host
            image = (image.float() / 2 + 0.5).clamp(0, 1)
            if output_type in ("np", "pil"):
                image = image.permute(0, 2, 3, 1).cpu().numpy()
                if output_type == "pil":
                    from PIL import Image
                    image = [Image.fromarray((im * 255).round().astype("uint8")) for im in image]
    return _Output(image) if return_dict else (image,)


class Generator4Embeds:
    """Reference wrapper (custom_pipeline.py:456-492; low-level variant custom_pipeline_low_level.py:571-616): sdxl-turbo + IP-Adapter, guidance 0,
    the EEG-predicted image embedding as the only condition.  `pipe`: a StandInSDXLPipeline (offline).  Without it the real diffusers pipeline is
    built, which needs `diffusers` and the stabilityai/sdxl-turbo + h94/IP-Adapter checkpoints -- absent offline, so that path raises here."""

    def __init__(self, num_inference_steps=1, device='cuda', img2img_strength=1, low_level_image=None, low_level_latent=None, pipe=None):
        self.num_inference_steps = num_inference_steps
        self.device = device
        self.img2img_strength, self.low_level_image, self.low_level_latent = img2img_strength, low_level_image, low_level_latent
        if pipe is not None:
            self.pipe, self.dtype, self._stand_in = pipe, pipe.dtype, True
            return
        try:
            from diffusers import DiffusionPipeline
        except ImportError as e:
            raise EegclipError("Generator4Embeds() builds stabilityai/sdxl-turbo + h94/IP-Adapter through `diffusers`; neither the package nor the "
                               "checkpoints exist in this offline build.  Pass pipe=StandInSDXLPipeline(...) to run the sampling loop on the SDXL-shaped "
                               "stand-in UNet (latents out), or install diffusers + the checkpoints") from e
        self.dtype, self._stand_in = torch.float16, False
        pipe = DiffusionPipeline.from_pretrained("stabilityai/sdxl-turbo", torch_dtype=torch.float16, variant="fp16")
        pipe.to(device)
        pipe.load_ip_adapter("h94/IP-Adapter", subfolder="sdxl_models", weight_name="ip-adapter_sdxl_vit-h.safetensors", torch_dtype=torch.float16)
        pipe.set_ip_adapter_scale(1)
        install_cross_attention_processors(pipe.unet, scale=1.0)
        self.pipe = pipe

    def generate(self, image_embeds, text_prompt='', generator=None):
        image_embeds = image_embeds.to(device=self.device, dtype=self.dtype)
        if image_embeds.dim() == 1:
            image_embeds = image_embeds[None]
        if not self._stand_in:
            return self.pipe(prompt=text_prompt, ip_adapter_image_embeds=[image_embeds.unsqueeze(1)], num_inference_steps=self.num_inference_steps,
                             guidance_scale=0.0, generator=generator).images[0]
        return self.pipe.generate_ip_adapter_embeds(prompt=text_prompt, ip_adapter_embeds=image_embeds, num_inference_steps=self.num_inference_steps,
                                                    guidance_scale=0.0, generator=generator, img2img_strength=self.img2img_strength,
                                                    low_level_image=self.low_level_image, low_level_latent=self.low_level_latent).images[0]


def bench_sampling_loop(images=8, steps=50, latent=128, guidance_scale=5.0, dtype=torch.float16):
    """BASELINE configs[4]: 50-step DDIM sampling of `images` images per GPU with classifier-free guidance on the SDXL-shaped stand-in (1024 px:
    128 x 128 latents, 4096 / 1024 tokens).  Returns a dict for bench.py's `secondary` object."""
    import time
    pipe = StandInSDXLPipeline(SDXLShapedUNet(dtype=dtype), DDIMScheduler(), device="cuda", dtype=dtype, default_sample_size=latent)
    emb = torch.randn(images, 1024, device="cuda", dtype=dtype)
    gen = torch.Generator(device="cuda").manual_seed(0)
    pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=2, guidance_scale=guidance_scale, generator=gen)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=steps, guidance_scale=guidance_scale, generator=gen).images
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_layers = sum(pipe.unet.stage_layers)
    B = 2 * images
    flops = 0.0
    for d, n, tok in zip(pipe.unet.stage_dims, pipe.unet.stage_layers, [(latent // 2) ** 2, (latent // 4) ** 2, (latent // 4) ** 2, (latent // 4) ** 2, (latent // 2) ** 2]):
        flops += n * (2 * 2.0 * B * tok * d * d + 4.0 * B * tok * 81 * d)            # to_q + to_out GEMMs, QK^T + PV over 77 + 4 tokens
    return {"stand_in": True, "workload": f"configs[4] shape: {steps}-step DDIM, {images} images x CFG pair, {latent * 8} px ({latent}x{latent} latents), SDXL-SHAPED STAND-IN "
                        f"UNet ({n_layers} cross-attention positions with IP-Adapter branch; no self-attention / ResNets / trained weights)",
            "steps": steps, "seconds": round(dt, 3), "ms_per_step": round(1e3 * dt / steps, 2), "images_per_s": round(images / dt, 2),
            "attention_stack_TFLOPs": round(flops * steps / dt / 1e12, 1), "finite": bool(torch.isfinite(out.float()).all())}
