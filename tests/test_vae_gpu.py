"""SURVEY section 8 row (f3), the VAE ends of the low-level reconstruction path (custom_pipeline_low_level.py:8-31 encode, custom_pipeline.py:421 decode) on the
MI355X: csrc/vae.hip layer by layer against torch fp64 / fp32 functional ops, and eeg_image_decode_amd.vae.SDXLShapedVAE (SDXL's VAE layout, random weights:
diffusers and its checkpoints are absent offline -- parity unpinned, oracle/sdxl_vae.py restates the published architecture) end to end in bf16 against the
fp32 restatement at north_star's bf16 budget (1e-2 of the output scale)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from eeg_image_decode_amd import _abi

pytestmark = pytest.mark.gpu


def _frame(t, pad):
    """(N, C, H, W) -> padded NHWC"""
    N, C, H, W = t.shape
    f = torch.zeros(N, H + 2 * pad, W + 2 * pad, C, dtype=t.dtype, device=t.device)
    f[:, pad:pad + H, pad:pad + W, :] = t.permute(0, 2, 3, 1)
    return f


def _unframe(f, pad):
    return f[:, pad:f.shape[1] - pad, pad:f.shape[2] - pad, :].permute(0, 3, 1, 2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [
    dict(Cin=128, Cout=128, H=16, W=24),                                   # ResNet convolution on the matrix cores
    dict(Cin=256, Cout=128, H=9, W=7, res=1),                              # odd sizes (a partial 128-pixel tile), residual epilogue
    dict(Cin=128, Cout=256, H=8, W=8, KS=1),                               # 1 x 1 shortcut
    dict(Cin=128, Cout=128, H=8, W=12, up=1),                              # Upsample2D + conv without the upsampled tensor
    dict(Cin=128, Cout=128, H=16, W=16, stride=2),                         # the encoder's downsampler: pad (0, 1, 0, 1), stride 2
    dict(Cin=4, Cout=512, H=8, W=8),                                       # decoder conv_in (direct kernel)
    dict(Cin=128, Cout=3, H=10, W=6, out_pad=0),                           # decoder conv_out
    dict(Cin=3, Cout=128, H=8, W=8),
    dict(Cin=8, Cout=8, H=5, W=5, KS=1, in_pad=1, out_pad=0),              # quant_conv
])
def test_conv16_against_torch_conv2d(dtype, case):
    from eeg_image_decode_amd._lib import lib, raw_stream
    torch.manual_seed(1)
    Cin, Cout, H, W = case["Cin"], case["Cout"], case["H"], case["W"]
    KS, stride, up = case.get("KS", 3), case.get("stride", 1), case.get("up", 0)
    in_pad, out_pad = case.get("in_pad", 1), case.get("out_pad", 1)
    N = 2
    x = torch.randn(N, Cin, H, W, device="cuda").to(dtype)
    w = (torch.randn(Cout, Cin, KS, KS, device="cuda") / (Cin * KS * KS) ** 0.5).to(dtype)
    b = torch.randn(Cout, device="cuda").to(dtype)
    xr, wr, br = x.double(), w.double(), b.double()
    if up:
        ref = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, br, padding=1)
        pads = (1, 1)
    elif stride == 2:
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, br, stride=2)
        pads = (0, 0)
    else:
        ref = F.conv2d(xr, wr, br, padding=(KS - 1) // 2)
        pads = ((KS - 1) // 2,) * 2
    Ho, Wo = ref.shape[2], ref.shape[3]
    res = torch.randn(N, Cout, Ho, Wo, device="cuda").to(dtype) if case.get("res") else None
    if res is not None:
        ref = ref + res.double()
    xin = _frame(x, in_pad)
    out = torch.full((N, Ho + 2 * out_pad, Wo + 2 * out_pad, Cout), float("nan"), dtype=dtype, device="cuda")
    rf = _frame(res, out_pad) if res is not None else None
    wp = w.permute(0, 2, 3, 1).reshape(Cout, KS * KS, Cin).contiguous()
    d = _abi.Conv16Desc(in_=xin.data_ptr(), W=wp.data_ptr(), out=out.data_ptr(), bias=b.data_ptr(), residual=rf.data_ptr() if rf is not None else None,
                        N=N, Hi=H, Wi=W, Cin=Cin, in_pad=in_pad, Ho=Ho, Wo=Wo, Cout=Cout, out_pad=out_pad, KS=KS, stride=stride, pad_top=pads[0],
                        pad_left=pads[1], upsample=up, dtype=_abi.DT_BF16 if dtype == torch.bfloat16 else _abi.DT_F16)
    assert lib().eegclip_conv16(d, raw_stream()) == 0
    torch.cuda.synchronize()
    got = _unframe(out, out_pad).double()
    tol = (8e-3 if dtype == torch.bfloat16 else 1.5e-3) * max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), atol=tol)
    if out_pad:                                                           # the border is never written
        assert torch.isnan(out[:, 0]).all() and torch.isnan(out[:, :, 0]).all() and torch.isnan(out[:, -1]).all() and torch.isnan(out[:, :, -1]).all()


@pytest.mark.parametrize("C,silu", [(128, 1), (512, 0)])
def test_groupnorm16_and_row_softmax(C, silu):
    from eeg_image_decode_amd._lib import lib, raw_stream
    torch.manual_seed(2)
    N, H, W = 2, 9, 13
    x = (torch.randn(N, C, H, W, device="cuda") * 2 + 0.5).bfloat16()
    g, b = (1 + 0.2 * torch.randn(C, device="cuda")).bfloat16(), (0.2 * torch.randn(C, device="cuda")).bfloat16()
    xin = _frame(x, 1)
    y = torch.zeros(N, H + 2, W + 2, C, dtype=torch.bfloat16, device="cuda")
    sums = torch.empty(N * 32 * 2, dtype=torch.float64, device="cuda")
    assert lib().eegclip_groupnorm16(xin.data_ptr(), N, H, W, C, 1, 32, g.data_ptr(), b.data_ptr(), 1e-6, silu, y.data_ptr(), 1, sums.data_ptr(), _abi.DT_BF16,
                                     raw_stream()) == 0
    ref = F.group_norm(x.double(), 32, g.double(), b.double(), 1e-6)
    if silu:
        ref = F.silu(ref)
    np.testing.assert_allclose(_unframe(y, 1).double().cpu().numpy(), ref.cpu().numpy(), atol=2e-2)
    assert float(y[:, 0].abs().max()) == 0.0 and float(y[:, :, -1].abs().max()) == 0.0
    s = (torch.randn(70, 384, device="cuda") * 3).bfloat16()
    want = torch.softmax(0.25 * s.double(), dim=-1)
    assert lib().eegclip_softmax_rows16(s.data_ptr(), 70, 384, 384, 0.25, _abi.DT_BF16, raw_stream()) == 0
    np.testing.assert_allclose(s.double().cpu().numpy(), want.cpu().numpy(), atol=4e-3)


def _state(vae):
    return {k: v.detach().float().cpu() for k, v in vae.state_dict().items()}


def test_vae_state_dict_has_autoencoderkl_keys():
    from eeg_image_decode_amd.vae import SDXLShapedVAE
    vae = SDXLShapedVAE()
    keys = set(vae.state_dict())
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.resnets.1.norm2.bias", "encoder.down_blocks.2.downsamplers.0.conv.weight",
              "encoder.down_blocks.1.resnets.0.conv_shortcut.weight", "encoder.mid_block.attentions.0.to_out.0.bias", "encoder.mid_block.attentions.0.group_norm.weight",
              "encoder.conv_norm_out.weight", "encoder.conv_out.bias", "quant_conv.weight", "post_quant_conv.bias", "decoder.conv_in.weight",
              "decoder.mid_block.resnets.1.conv2.weight", "decoder.up_blocks.0.resnets.2.conv1.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.bias",
              "decoder.up_blocks.2.upsamplers.0.conv.weight", "decoder.conv_norm_out.bias", "decoder.conv_out.weight"):
        assert k in keys, k
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in keys and "encoder.down_blocks.3.downsamplers.0.conv.weight" not in keys
    sd = vae.state_dict()
    assert tuple(sd["encoder.conv_out.weight"].shape) == (8, 512, 3, 3) and tuple(sd["decoder.up_blocks.3.resnets.0.conv_shortcut.weight"].shape) == (128, 256, 1, 1)
    assert sum(p.numel() for p in vae.parameters()) == 83_653_863           # the published parameter count of the SDXL VAE (AutoencoderKL, this config)


def test_vae_decode_and_encode_match_the_fp32_restatement():
    """bf16 HIP path vs oracle/sdxl_vae.py (fp32, same weights after the bf16 rounding of the parameters): 16 x 16 latents -> 128 x 128 image and back"""
    from eeg_image_decode_amd.vae import SDXLShapedVAE
    from oracle import sdxl_vae as ovae
    vae = SDXLShapedVAE(seed=3).cuda()
    P = _state(vae)
    torch.manual_seed(4)
    z = torch.randn(2, 4, 16, 16)
    img = vae.decode(z.cuda().bfloat16()).float().cpu()
    ref = ovae.decode(P, z.bfloat16().float())
    assert img.shape == (2, 3, 128, 128)
    scale = float(ref.abs().max())
    err = (img - ref).abs()
    assert float(err.max()) <= 3e-2 * scale and float(err.mean()) <= 4e-3 * scale, (float(err.max()) / scale, float(err.mean()) / scale)
    # a second call reuses the recycled frames (their zero borders must have survived)
    img2 = vae.decode(z.cuda().bfloat16()).float().cpu()
    assert torch.equal(img, img2)
    x = torch.randn(2, 3, 128, 128)
    mom = vae.encode_moments(x.cuda().bfloat16()).float().cpu().permute(0, 3, 1, 2)
    refm = ovae.encode_moments(P, x.bfloat16().float())
    scale = float(refm.abs().max())
    err = (mom - refm).abs()
    assert float(err.max()) <= 3e-2 * scale and float(err.mean()) <= 4e-3 * scale, (float(err.max()) / scale, float(err.mean()) / scale)
    # latent_dist.sample(generator): the reference's noise, drawn in its shape and order
    g = torch.Generator(device="cuda").manual_seed(11)
    lat = vae.encode(x.cuda().bfloat16(), generator=g).float().cpu()
    g2 = torch.Generator(device="cuda").manual_seed(11)
    noise = torch.randn((2, 4, 16, 16), generator=g2, device="cuda", dtype=torch.bfloat16).float().cpu()
    want = ovae.sample(mom, noise)
    np.testing.assert_allclose(lat.numpy(), want.numpy(), atol=2e-2 * max(1.0, float(want.abs().max())))
    np.testing.assert_allclose(vae.encode(x.cuda().bfloat16(), sample=False).float().cpu().numpy(), ovae.sample(mom).numpy(), atol=1e-2 * max(1.0, scale))


def test_low_level_image_pipeline_encodes_with_the_vae_and_decodes_the_result():
    """the whole low-level path with its VAE ends (custom_pipeline_low_level.py:8-31, 331-389 + custom_pipeline.py:421): low_level_image -> vae.encode ->
    * scaling_factor + noise -> img2img sampling on the stand-in UNet -> / scaling_factor -> vae.decode -> postprocess.  Pieces against the restatement: the
    start latent the loop sees and the image decoded from the loop's final latents."""
    from eeg_image_decode_amd.sdxl import DDIMScheduler, SDXLShapedUNet, StandInSDXLPipeline
    from eeg_image_decode_amd.vae import SDXLShapedVAE
    from oracle import sdxl_vae as ovae
    dtype = torch.bfloat16
    vae = SDXLShapedVAE(seed=3, dtype=dtype)
    pipe = StandInSDXLPipeline(SDXLShapedUNet(stage_layers=(1, 1, 1, 1, 1), dtype=dtype, seed=5), DDIMScheduler(), device="cuda", dtype=dtype, default_sample_size=16,
                               vae=vae)
    P = _state(pipe.vae)
    emb = torch.randn(1, 1024, generator=torch.Generator().manual_seed(4)).cuda().to(dtype)
    low = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(5)) * 2 - 1
    seen = {}
    real = pipe.prepare_latents_latent2img
    pipe.prepare_latents_latent2img = lambda lat, *a, **k: (seen.setdefault("enc", lat.float().cpu()), real(lat, *a, **k))[1]
    lat = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=4, guidance_scale=0.0, img2img_strength=0.5,
                                          low_level_image=low, generator=torch.Generator(device="cuda").manual_seed(12), output_type="latent").images
    # (1) what the encoder handed to the loop = latent_dist.sample with the SAME generator stream: the encoder's noise is the generator's first draw
    g = torch.Generator(device="cuda").manual_seed(12)
    noise = torch.randn((1, 4, 16, 16), generator=g, device="cuda", dtype=dtype).float().cpu()
    want = ovae.sample(ovae.encode_moments(P, low.to(dtype).float()), noise)
    np.testing.assert_allclose(seen["enc"].numpy(), want.numpy(), atol=3e-2 * max(1.0, float(want.abs().max())))
    # (2) the image: same call with output_type "pt" = decode(latents / scaling_factor) / 2 + 0.5 clamped
    img = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=4, guidance_scale=0.0, img2img_strength=0.5,
                                          low_level_image=low, generator=torch.Generator(device="cuda").manual_seed(12), output_type="pt").images
    assert img.shape == (1, 3, 128, 128) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    ref = (ovae.decode(P, (lat.float().cpu() / pipe.vae.scaling_factor).to(dtype).float()) / 2 + 0.5).clamp(0, 1)
    err = (img.cpu() - ref).abs()
    assert float(err.max()) <= 3e-2 and float(err.mean()) <= 4e-3, (float(err.max()), float(err.mean()))
    pil = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=2, guidance_scale=0.0, output_type="pil").images
    assert pil[0].size == (128, 128)
