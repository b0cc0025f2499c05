"""SDXL VAE (diffusers' AutoencoderKL) on HIP kernels: the `vae.encode` of the low-level reconstruction start (Generation/custom_pipeline_low_level.py:8-31:
latents = vae.encode(image).latent_dist.sample(generator) * scaling_factor) and the `vae.decode` that ends every sampling loop
(Generation/custom_pipeline.py:421: image = vae.decode(latents / scaling_factor)).

The reference takes both from the `stabilityai/sdxl-turbo` checkpoint through diffusers 0.30.0 -- neither is available offline -- so, like `sdxl.SDXLShapedUNet`,
`SDXLShapedVAE` is the module with SDXL's VAE LAYOUT (block_out_channels 128 / 256 / 512 / 512, two ResNet blocks per encoder level and three per decoder
level, GroupNorm(32), one single-head self-attention in each mid block, 4 latent channels, scaling_factor 0.13025) and the state_dict keys of AutoencoderKL,
so that a real checkpoint loads with `load_state_dict`; offline its weights are random.  Parity is therefore held against a restatement of the published
architecture (oracle/sdxl_vae.py, fp32 torch; "parity unpinned" like every diffusers row of SURVEY.md section 8c).

Arithmetic: csrc/vae.hip -- 16-bit padded-NHWC activations, fp32 accumulation; 3 x 3 / 1 x 1 convolutions as implicit GEMMs on the matrix cores (nearest-2x
upsampling and stride-2 downsampling folded into the row addressing), GroupNorm + SiLU as one statistics and one apply pass, the mid-block attention as
three eegclip_gemm16 launches around a row softmax.  The nn.Conv2d / nn.GroupNorm / nn.Linear children hold parameters only; they are never called.
bf16 by default: the reference upcasts its fp16 VAE to fp32 for the decode because fp16 overflows there (custom_pipeline.py:412-419); bf16 has fp32's range.
"""
import math

import torch
import torch.nn as nn

from . import _abi
from ._lib import EegclipError, check, lib, raw_stream, require_cuda

GROUPS, EPS = 32, 1e-6


def _dt(dtype):
    if dtype == torch.bfloat16:
        return _abi.DT_BF16
    if dtype == torch.float16:
        return _abi.DT_F16
    raise EegclipError("the VAE kernels run in bf16 or fp16")


class _Resnet(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(GROUPS, cin, eps=EPS)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(GROUPS, cout, eps=EPS)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(GROUPS, c, eps=EPS)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])


class _Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([_Attention(c)])
        self.resnets = nn.ModuleList([_Resnet(c, c), _Resnet(c, c)])


class _Sampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)


class _Level(nn.Module):
    def __init__(self, cin, cout, n_res, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout) for i in range(n_res)])
        if down:
            self.downsamplers = nn.ModuleList([_Sampler(cout)])
        if up:
            self.upsamplers = nn.ModuleList([_Sampler(cout)])


class _Encoder(nn.Module):
    def __init__(self, chans, layers, latent):
        super().__init__()
        self.conv_in = nn.Conv2d(3, chans[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([_Level(chans[max(i - 1, 0)], c, layers, down=i < len(chans) - 1) for i, c in enumerate(chans)])
        self.mid_block = _Mid(chans[-1])
        self.conv_norm_out = nn.GroupNorm(GROUPS, chans[-1], eps=EPS)
        self.conv_out = nn.Conv2d(chans[-1], 2 * latent, 3, padding=1)


class _Decoder(nn.Module):
    def __init__(self, chans, layers, latent):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0])
        self.up_blocks = nn.ModuleList([_Level(rev[max(i - 1, 0)], c, layers + 1, up=i < len(rev) - 1) for i, c in enumerate(rev)])
        self.conv_norm_out = nn.GroupNorm(GROUPS, rev[-1], eps=EPS)
        self.conv_out = nn.Conv2d(rev[-1], 3, 3, padding=1)


class SDXLShapedVAE(nn.Module):
    def __init__(self, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, scaling_factor=0.13025, dtype=torch.bfloat16,
                 seed=0):
        super().__init__()
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)
            self.encoder = _Encoder(block_out_channels, layers_per_block, latent_channels)
            self.decoder = _Decoder(block_out_channels, layers_per_block, latent_channels)
            self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
            self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.latent_channels, self.scaling_factor = latent_channels, scaling_factor
        self.downscale = 2 ** (len(block_out_channels) - 1)
        self.to(dtype)
        for p in self.parameters():
            p.requires_grad_(False)
        self._packed, self._pool, self._sums = {}, {}, None

    def forward(self, *a, **k):
        raise EegclipError("SDXLShapedVAE holds parameters; call .encode(image) / .decode(latents) (HIP kernels). There is no eager path.")

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    # ---- plumbing: packed weights, padded frames ---------------------------------------------------------------------------------------------------
    def _w(self, mod):
        """the layer's weight as [Cout][KS * KS][Cin] (what csrc/vae.hip contracts over), packed once per parameter version"""
        w = mod.weight
        key = (id(w), w._version, w.data_ptr())
        hit = self._packed.get(id(mod))
        if hit is None or hit[0] != key:
            p = w.detach()
            p = (p.permute(0, 2, 3, 1).reshape(p.shape[0], -1, p.shape[1]) if p.dim() == 4 else p.reshape(p.shape[0], 1, p.shape[1])).contiguous()
            hit = self._packed[id(mod)] = (key, p)
        return hit[1]

    def _frame(self, N, H, W, C, pad):
        """a (N, H + 2 pad, W + 2 pad, C) tensor whose border is zero: borders are never written by the kernels, so frames are recycled without clearing"""
        shape = (N, H + 2 * pad, W + 2 * pad, C)
        free = self._pool.setdefault((shape, pad), [])        # (keyed by the padding too: an unpadded frame of the same shape is written edge to edge)
        if free:
            return free.pop()
        f = torch.zeros(shape, dtype=self.dtype, device=self.device)
        f._eegclip_pad = pad
        return f

    def _done(self, *frames):
        for f in frames:
            self._pool.setdefault((tuple(f.shape), f._eegclip_pad), []).append(f)

    def _conv(self, x, xpad, mod, out_pad=1, stride=1, pads=None, upsample=False, residual=None, KS=None):
        """x: (N, Hi + 2 xpad, Wi + 2 xpad, Cin) frame -> (N, Ho + 2 out_pad, Wo + 2 out_pad, Cout) frame.  pads = (top, left, bottom, right) zero padding of
        the convolution (default: "same"); upsample: over the nearest-2x upsampled input"""
        N, Hi, Wi, Cin = x.shape[0], x.shape[1] - 2 * xpad, x.shape[2] - 2 * xpad, x.shape[3]
        w = self._w(mod)
        Cout, KS = w.shape[0], (KS or int(round(math.sqrt(w.shape[1]))))
        pt, pleft, pb, pr = pads if pads is not None else ((KS - 1) // 2,) * 4
        if upsample:
            Ho, Wo = 2 * Hi, 2 * Wi
        else:
            Ho, Wo = (Hi + pt + pb - KS) // stride + 1, (Wi + pleft + pr - KS) // stride + 1
        out = self._frame(N, Ho, Wo, Cout, out_pad)
        d = _abi.Conv16Desc(in_=x.data_ptr(), W=w.data_ptr(), out=out.data_ptr(), bias=mod.bias.data_ptr() if mod.bias is not None else None,
                            residual=residual.data_ptr() if residual is not None else None, N=N, Hi=Hi, Wi=Wi, Cin=Cin, in_pad=xpad, Ho=Ho, Wo=Wo, Cout=Cout,
                            out_pad=out_pad, KS=KS, stride=stride, pad_top=pt, pad_left=pleft, upsample=int(upsample), dtype=_dt(self.dtype))
        check(lib().eegclip_conv16(d, raw_stream()), "conv16")
        return out

    def _gn(self, x, xpad, mod, silu=True, out_pad=1):
        N, H, W, C = x.shape[0], x.shape[1] - 2 * xpad, x.shape[2] - 2 * xpad, x.shape[3]
        if self._sums is None or self._sums.numel() < N * GROUPS * 2:
            self._sums = torch.empty(N * GROUPS * 2, dtype=torch.float64, device=self.device)
        y = self._frame(N, H, W, C, out_pad)
        check(lib().eegclip_groupnorm16(x.data_ptr(), N, H, W, C, xpad, GROUPS, mod.weight.data_ptr(), mod.bias.data_ptr(), float(mod.eps), int(silu),
                                        y.data_ptr(), out_pad, self._sums.data_ptr(), _dt(self.dtype), raw_stream()), "groupnorm16")
        return y

    # ---- blocks (diffusers ResnetBlock2D / Attention / UNetMidBlock2D with one attention, as AutoencoderKL configures them) -------------------------
    def _resnet(self, x, r):
        h = self._gn(x, 1, r.norm1)
        h2 = self._conv(h, 1, r.conv1)
        self._done(h)
        h = self._gn(h2, 1, r.norm2)
        self._done(h2)
        sc = self._conv(x, 1, r.conv_shortcut, KS=1) if hasattr(r, "conv_shortcut") else x
        out = self._conv(h, 1, r.conv2, residual=sc)
        self._done(h)
        if sc is not x:
            self._done(sc)
        return out

    def _attention(self, x, a):
        """single-head self-attention over the H W positions (head dim = C), residual connection; x: padded frame"""
        N, H, W, C = x.shape[0], x.shape[1] - 2, x.shape[2] - 2, x.shape[3]
        T = H * W
        if T % 128 or C % 128:
            raise EegclipError(f"the mid-block attention takes H * W and C multiples of 128 (got {T}, {C})")
        L, st, dt = lib(), raw_stream(), _dt(self.dtype)
        hn = self._gn(x, 1, a.group_norm, silu=False, out_pad=0)              # (N, H, W, C) = tokens (N, T, C)
        q = self._conv(hn, 0, a.to_q, out_pad=0, KS=1)
        k = self._conv(hn, 0, a.to_k, out_pad=0, KS=1)
        o = self._frame(N, H, W, C, 0)
        wv = a.to_v.weight
        vt = torch.empty(C, T, dtype=self.dtype, device=self.device)
        s = torch.empty(T, T, dtype=self.dtype, device=self.device)
        for n in range(N):
            tok = hn[n].reshape(T, C)
            # v^T = Wv tokens^T (the value bias is added behind the softmax-weighted sum: the weights of a row add up to one)
            check(L.eegclip_gemm16(wv.data_ptr(), C, tok.data_ptr(), C, vt.data_ptr(), T, None, None, 0, 0, C, T, C, dt, st), "gemm16 v^T")
            check(L.eegclip_gemm16(q[n].data_ptr(), C, k[n].data_ptr(), C, s.data_ptr(), T, None, None, 0, 0, T, T, C, dt, st), "gemm16 q k^T")
            check(L.eegclip_softmax_rows16(s.data_ptr(), T, T, T, 1.0 / math.sqrt(C), dt, st), "softmax_rows16")
            check(L.eegclip_gemm16(s.data_ptr(), T, vt.data_ptr(), T, o[n].data_ptr(), C, a.to_v.bias.data_ptr(), None, 0, 0, T, C, T, dt, st), "gemm16 p v")
        out = self._conv(o, 0, a.to_out[0], out_pad=1, residual=x, KS=1)
        self._done(hn, q, k, o)
        return out

    def _mid(self, x, m):
        h = self._resnet(x, m.resnets[0])
        self._done(x)
        h2 = self._attention(h, m.attentions[0])
        self._done(h)
        h = self._resnet(h2, m.resnets[1])
        self._done(h2)
        return h

    def _to_frame(self, t):
        """(N, C, H, W) -> padded NHWC frame (a layout change at the boundary; torch plumbing)"""
        N, C, H, W = t.shape
        f = self._frame(N, H, W, C, 1)
        f[:, 1:-1, 1:-1, :] = t.permute(0, 2, 3, 1)
        return f

    # ---- the two calls of the reference -----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, latents):
        """latents (N, 4, h, w) -> image (N, 3, 8 h, 8 w)      (AutoencoderKL.decode: post_quant_conv, Decoder)"""
        require_cuda(latents, "latents")
        if latents.dim() != 4 or latents.shape[1] != self.latent_channels:
            raise EegclipError(f"latents must be (N, {self.latent_channels}, h, w); got {tuple(latents.shape)}")
        d = self.decoder
        x0 = self._to_frame(latents.to(self.dtype))
        x = self._conv(x0, 1, self.post_quant_conv, KS=1)
        self._done(x0)
        h = self._conv(x, 1, d.conv_in)
        self._done(x)
        h = self._mid(h, d.mid_block)
        for lvl in d.up_blocks:
            for r in lvl.resnets:
                h2 = self._resnet(h, r)
                self._done(h)
                h = h2
            if hasattr(lvl, "upsamplers"):
                h2 = self._conv(h, 1, lvl.upsamplers[0].conv, upsample=True)
                self._done(h)
                h = h2
        hn = self._gn(h, 1, d.conv_norm_out)
        self._done(h)
        img = self._conv(hn, 1, d.conv_out, out_pad=0)
        self._done(hn)
        out = img.permute(0, 3, 1, 2).contiguous()
        self._done(img)
        return out

    @torch.no_grad()
    def encode_moments(self, image):
        """image (N, 3, H, W) -> (N, H / 8, W / 8, 8) [mean | logvar] as the encoder + quant_conv leave them (NHWC)"""
        require_cuda(image, "image")
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] % self.downscale or image.shape[3] % self.downscale:
            raise EegclipError(f"image must be (N, 3, H, W) with H, W multiples of {self.downscale}; got {tuple(image.shape)}")
        e = self.encoder
        x = self._to_frame(image.to(self.dtype))
        h = self._conv(x, 1, e.conv_in)
        self._done(x)
        for lvl in e.down_blocks:
            for r in lvl.resnets:
                h2 = self._resnet(h, r)
                self._done(h)
                h = h2
            if hasattr(lvl, "downsamplers"):                 # Downsample2D in the VAE: pad (0, 1, 0, 1) + 3 x 3 stride-2 convolution without padding
                h2 = self._conv(h, 1, lvl.downsamplers[0].conv, stride=2, pads=(0, 0, 1, 1))
                self._done(h)
                h = h2
        h = self._mid(h, e.mid_block)
        hn = self._gn(h, 1, e.conv_norm_out)
        self._done(h)
        m = self._conv(hn, 1, e.conv_out)
        self._done(hn)
        mom = self._conv(m, 1, self.quant_conv, out_pad=0, KS=1)
        self._done(m)
        return mom

    @torch.no_grad()
    def encode(self, image, generator=None, sample=True):
        """latents = AutoencoderKL.encode(image).latent_dist.sample(generator) (or .mode() with sample=False): (N, 4, H / 8, W / 8).  The noise is drawn
        with torch.randn in the reference's shape and order (NCHW) so that a shared generator reproduces it."""
        mom = self.encode_moments(image)
        N, h, w, _ = mom.shape
        Lc = self.latent_channels
        noise = None
        if sample:
            noise = torch.randn((N, Lc, h, w), generator=generator, device=mom.device, dtype=mom.dtype).permute(0, 2, 3, 1).contiguous()
        z = torch.empty(N, h, w, Lc, dtype=mom.dtype, device=mom.device)
        check(lib().eegclip_vae_sample16(mom.data_ptr(), noise.data_ptr() if noise is not None else None, z.data_ptr(), N * h * w, Lc, _dt(self.dtype),
                                         raw_stream()), "vae_sample16")
        self._done(mom)
        return z.permute(0, 3, 1, 2).contiguous()


def bench_decode(images=1, latent=128, dtype=torch.bfloat16, reps=3):
    """ms per decode of `images` latents of latent x latent (128 -> 1024 x 1024 pixels) + the algorithmic flops of its convolutions and attention"""
    vae = SDXLShapedVAE(dtype=dtype).cuda()
    z = torch.randn(images, 4, latent, latent, device="cuda", dtype=dtype)
    vae.decode(z)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        vae.decode(z)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # 2 * 9 * Cin * Cout flops per output pixel of every 3 x 3 convolution (the 1 x 1 / attention terms are < 3 %)
    res, fl = latent, 0.0
    chans = [512, 512, 256, 128]
    fl += 2 * 9 * 4 * 512 * res * res + 4 * 2 * 9 * 512 * 512 * res * res + 4 * res * res * res * res * 512
    cin = 512
    for i, c in enumerate(chans):
        fl += 2 * 9 * res * res * (cin * c + 5 * c * c) + (2 * cin * c * res * res if cin != c else 0)
        if i < 3:
            res *= 2
            fl += 2 * 9 * c * c * res * res
        cin = c
    fl += 2 * 9 * 128 * 3 * res * res
    return {"images": images, "pixels": f"{res}x{res}", "dtype": str(dtype).split(".")[-1], "ms_per_decode": round(ms, 3),
            "algorithmic_TFLOPs": round(images * fl / ms / 1e9, 1), "stand_in": True}
