"""TEST INFRASTRUCTURE (oracle): fp32 torch restatement of the SDXL VAE's published architecture -- diffusers 0.30.0 `AutoencoderKL` with the SDXL config
(block_out_channels 128 / 256 / 512 / 512, layers_per_block 2, norm_num_groups 32, eps 1e-6, one single-head attention per mid block, latent_channels 4) --
over a state_dict with AutoencoderKL's keys.  Follows the call sites Generation/custom_pipeline_low_level.py:8-31 (`vae.encode(image).latent_dist.sample`)
and Generation/custom_pipeline.py:421 (`vae.decode`).

PARITY UNPINNED: diffusers is neither vendored by the reference nor installed here and no checkpoint is available offline; this file
restates Encoder / Decoder / ResnetBlock2D / Attention / Downsample2D / Upsample2D / DiagonalGaussianDistribution from the published algorithm.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import it."""
import torch
import torch.nn.functional as F

GROUPS, EPS = 32, 1e-6


def _gn(P, pre, x):
    return F.group_norm(x, GROUPS, P[pre + ".weight"], P[pre + ".bias"], EPS)


def _conv(P, pre, x, stride=1, padding=1):
    return F.conv2d(x, P[pre + ".weight"], P[pre + ".bias"], stride=stride, padding=padding)


def resnet(P, pre, x):
    """ResnetBlock2D (no time embedding in the VAE): norm1 -> SiLU -> conv1 -> norm2 -> SiLU -> conv2, + (1x1 conv) shortcut"""
    h = _conv(P, pre + ".conv1", F.silu(_gn(P, pre + ".norm1", x)))
    h = _conv(P, pre + ".conv2", F.silu(_gn(P, pre + ".norm2", h)))
    sc = _conv(P, pre + ".conv_shortcut", x, padding=0) if pre + ".conv_shortcut.weight" in P else x
    return sc + h


def attention(P, pre, x):
    """Attention(heads = 1, dim_head = C, residual_connection, group norm): softmax(q k^T / sqrt(C)) v over the H W positions"""
    N, C, H, W = x.shape
    t = _gn(P, pre + ".group_norm", x).reshape(N, C, H * W).transpose(1, 2)
    q = F.linear(t, P[pre + ".to_q.weight"], P[pre + ".to_q.bias"])
    k = F.linear(t, P[pre + ".to_k.weight"], P[pre + ".to_k.bias"])
    v = F.linear(t, P[pre + ".to_v.weight"], P[pre + ".to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) / C ** 0.5, dim=-1) @ v
    o = F.linear(a, P[pre + ".to_out.0.weight"], P[pre + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(N, C, H, W)


def mid(P, pre, x):
    x = resnet(P, pre + ".resnets.0", x)
    x = attention(P, pre + ".attentions.0", x)
    return resnet(P, pre + ".resnets.1", x)


def decode(P, z, levels=4, layers_per_block=2):
    """AutoencoderKL.decode: post_quant_conv, then Decoder (conv_in, mid block, up blocks with nearest-2x Upsample2D + conv, GroupNorm, SiLU, conv_out)"""
    x = _conv(P, "post_quant_conv", z, padding=0)
    x = _conv(P, "decoder.conv_in", x)
    x = mid(P, "decoder.mid_block", x)
    for i in range(levels):
        for j in range(layers_per_block + 1):
            x = resnet(P, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i < levels - 1:
            x = _conv(P, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    return _conv(P, "decoder.conv_out", F.silu(_gn(P, "decoder.conv_norm_out", x)))


def encode_moments(P, image, levels=4, layers_per_block=2):
    """Encoder + quant_conv: (N, 8, H / 8, W / 8) = [mean | logvar]; Downsample2D of the VAE pads (0, 1, 0, 1) and convolves with stride 2, no padding"""
    x = _conv(P, "encoder.conv_in", image)
    for i in range(levels):
        for j in range(layers_per_block):
            x = resnet(P, f"encoder.down_blocks.{i}.resnets.{j}", x)
        if i < levels - 1:
            x = _conv(P, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(x, (0, 1, 0, 1)), stride=2, padding=0)
    x = mid(P, "encoder.mid_block", x)
    x = _conv(P, "encoder.conv_out", F.silu(_gn(P, "encoder.conv_norm_out", x)))
    return _conv(P, "quant_conv", x, padding=0)


def sample(moments, noise=None):
    """DiagonalGaussianDistribution: mean + exp(0.5 clamp(logvar, -30, 20)) * noise (noise None: the mode)"""
    mean, logvar = moments.chunk(2, dim=1)
    if noise is None:
        return mean
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise
