"""Oracle: SDXL UNet cross-attention with the IP-Adapter branch (numpy float64).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED: the arithmetic lives in diffusers==0.30.0
(`models/attention_processor.py`: AttnProcessor2_0 / IPAdapterAttnProcessor2_0), which the reference calls at
Generation/custom_pipeline.py:365-373 but does not vendor, and diffusers is not installed here.  Restated from the published
algorithm:  heads of 64;  out = softmax(q k^T / sqrt(64)) v  +  scale * softmax(q k_ip^T / sqrt(64)) v_ip   (scale = 1,
custom_pipeline.py:476 `set_ip_adapter_scale(1)`); the non-"plus" IP-Adapter projects the (B,1024) image embedding to 4 tokens of
2048 with Linear(1024 -> 4*2048) + LayerNorm(2048).
"""
import numpy as np


def _softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)


def cross_attention(q, k, v, heads, k_ip=None, v_ip=None, ip_scale=1.0):
    """q (B,HW,C), k/v (B,S,C), k_ip/v_ip (B,S_ip,C) -> (B,HW,C) float64"""
    B, HW, C = q.shape
    d = C // heads

    def split(t):
        return t.astype(np.float64).reshape(t.shape[0], t.shape[1], heads, d).transpose(0, 2, 1, 3)

    qh = split(q)

    def attend(kk, vv):
        p = _softmax(qh @ split(kk).transpose(0, 1, 3, 2) / np.sqrt(d))
        return p @ split(vv)

    o = attend(k, v)
    if k_ip is not None:
        o = o + ip_scale * attend(k_ip, v_ip)
    return o.transpose(0, 2, 1, 3).reshape(B, HW, C)


def image_projection(image_embeds, w, b, ln_w, ln_b, n_tokens=4, eps=1e-5):
    """diffusers ImageProjection: (B,1024) -> (B,4,2048): Linear then LayerNorm over the last dim."""
    x = image_embeds.astype(np.float64) @ w.astype(np.float64).T + b
    x = x.reshape(image_embeds.shape[0], n_tokens, -1)
    mu = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * ln_w + ln_b
