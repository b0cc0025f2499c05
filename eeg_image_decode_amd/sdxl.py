"""SDXL sampling side of the path (Generation/custom_pipeline.py): the cross-attention (+ IP-Adapter image branch) of every
UNet transformer block runs in ONE hand-written HIP kernel (csrc/cross_attn.hip) instead of two SDPA calls + an add.

* cross_attention(...)                 functional form on (B, HW, heads*64) fp16 / bf16 CUDA tensors.
* HIPIPAdapterAttnProcessor            drop-in for diffusers' IPAdapterAttnProcessor2_0 / AttnProcessor2_0:
                                       `unet.set_attn_processor({name: HIPIPAdapterAttnProcessor(...)})`; same call signature
                                       `proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, ...)`.
                                       The q/k/v/out projections stay plain library GEMMs (attn.to_q ...); self-attention layers
                                       (encoder_hidden_states is None) are not this kernel's job and are rejected.
* Generator4Embeds                     the reference's wrapper class (custom_pipeline.py:456-492).  Needs `diffusers` + the
                                       sdxl-turbo / IP-Adapter checkpoints, none of which exist offline: constructing it without them
                                       raises.  (parity for this row is unpinned.)
"""
import torch
import torch.nn as nn

from . import _abi
from ._lib import EegclipError, check, lib, require_cuda


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
                                       float(ip_scale), _abi.DT_F16 if q.dtype == torch.float16 else _abi.DT_BF16,
                                       torch.cuda.current_stream().cuda_stream), "cross_attn_fwd")
    return out


class HIPIPAdapterAttnProcessor(nn.Module):
    """Cross-attention processor with an optional IP-Adapter branch (to_k_ip / to_v_ip: Linear(cross_attention_dim -> hidden_size),
    the non-"plus" adapter with 4 image tokens; scale 1 as in the reference, custom_pipeline.py:476)."""

    def __init__(self, hidden_size, cross_attention_dim=2048, num_tokens=4, scale=1.0, with_ip=True):
        super().__init__()
        self.hidden_size, self.cross_attention_dim, self.num_tokens, self.scale = hidden_size, cross_attention_dim, num_tokens, scale
        self.to_k_ip = nn.Linear(cross_attention_dim, hidden_size, bias=False) if with_ip else None
        self.to_v_ip = nn.Linear(cross_attention_dim, hidden_size, bias=False) if with_ip else None

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
        if getattr(attn, "norm_cross", None):
            encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
        q = attn.to_q(hidden_states)
        k = attn.to_k(encoder_hidden_states)
        v = attn.to_v(encoder_hidden_states)
        k_ip = v_ip = None
        if ip_tokens is not None and self.to_k_ip is not None:
            if ip_tokens.dim() == 4:                                          # (B, n_images, tokens, dim)
                ip_tokens = ip_tokens.flatten(1, 2)
            k_ip, v_ip = self.to_k_ip(ip_tokens.to(q.dtype)), self.to_v_ip(ip_tokens.to(q.dtype))
        out = cross_attention(q, k, v, attn.heads, k_ip, v_ip, self.scale)
        out = attn.to_out[0](out)
        out = attn.to_out[1](out)
        if shape4 is not None:
            out = out.transpose(-1, -2).reshape(shape4)
        if getattr(attn, "residual_connection", False):
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


class Generator4Embeds:
    """Reference wrapper (custom_pipeline.py:456-492): sdxl-turbo + IP-Adapter, fp16, guidance 0, image embedding as the only condition."""

    def __init__(self, num_inference_steps=1, device='cuda'):
        try:
            from diffusers import DiffusionPipeline
        except ImportError as e:
            raise EegclipError("Generator4Embeds needs `diffusers` and the stabilityai/sdxl-turbo + h94/IP-Adapter checkpoints, which are "
                               "not available in this offline build; the cross-attention kernel itself is exercised by "
                               "eeg_image_decode_amd.sdxl.cross_attention / HIPIPAdapterAttnProcessor") from e
        self.num_inference_steps = num_inference_steps
        self.dtype = torch.float16
        self.device = device
        pipe = DiffusionPipeline.from_pretrained("stabilityai/sdxl-turbo", torch_dtype=torch.float16, variant="fp16")
        pipe.to(device)
        pipe.load_ip_adapter("h94/IP-Adapter", subfolder="sdxl_models", weight_name="ip-adapter_sdxl_vit-h.safetensors", torch_dtype=torch.float16)
        pipe.set_ip_adapter_scale(1)
        install_cross_attention_processors(pipe.unet, scale=1.0)
        self.pipe = pipe

    def generate(self, image_embeds, text_prompt='', generator=None):
        image_embeds = image_embeds.to(device=self.device, dtype=self.dtype)
        return self.pipe(prompt=text_prompt, ip_adapter_image_embeds=[image_embeds.unsqueeze(1)], num_inference_steps=self.num_inference_steps,
                         guidance_scale=0.0, generator=generator).images[0]
