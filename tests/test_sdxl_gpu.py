"""Rows F1 / F2 / f3 on the MI355X: the cross-attention processor (diffusers call signature, every GEMM on csrc/gemm16.hip), the sampling loop
`generate_ip_adapter_embeds` with both schedulers, the img2img start and `Generator4Embeds`, against the numpy oracle (oracle/sdxl_pipeline.py:
the loop and the schedulers are restatements of diffusers 0.30.0 -- parity unpinned --, the UNet is the SDXL-shaped stand-in)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import sdxl_attn
from sdxl_common import FakeAttention, check_processor_kv_cache_never_serves_another_tensors_projections, oracle_loop, small_pipe

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("dim,heads", [(640, 10), (1280, 20)])
def test_ip_adapter_cross_attention_processor_issues_no_library_gemm(dtype, dim, heads, monkeypatch):
    from eeg_image_decode_amd import sdxl
    torch.manual_seed(0)
    B, HW, cross = 2, 256, 2048
    attn = FakeAttention(dim, cross, heads).cuda().to(dtype)
    proc = sdxl.HIPIPAdapterAttnProcessor(dim, cross, scale=1.0).cuda().to(dtype)
    hs = torch.randn(B, HW, dim, device="cuda", dtype=dtype)
    text = torch.randn(B, 77, cross, device="cuda", dtype=dtype)
    ip = torch.randn(B, 1, 4, cross, device="cuda", dtype=dtype)
    # oracle on 16-bit projections computed by torch BEFORE the library GEMMs are forbidden
    with torch.no_grad():
        q, k, v = attn.to_q(hs), attn.to_k(text), attn.to_v(text)
        kip, vip = proc.to_k_ip(ip.flatten(1, 2)), proc.to_v_ip(ip.flatten(1, 2))
        ref = sdxl_attn.cross_attention(*(t.float().cpu().numpy() for t in (q, k, v)), heads, kip.float().cpu().numpy(), vip.float().cpu().numpy(), 1.0)
        ref_out = attn.to_out[0](torch.tensor(ref, dtype=torch.float32).cuda().to(dtype))

    def forbidden(*a, **k):
        raise AssertionError("library GEMM on the cross-attention path")
    monkeypatch.setattr(torch.nn.functional, "linear", forbidden)
    monkeypatch.setattr(torch.nn.Linear, "forward", forbidden)
    calls = []
    real = sdxl.linear16
    monkeypatch.setattr(sdxl, "linear16", lambda *a, **k: (calls.append(a[1].shape), real(*a, **k))[1])
    with torch.no_grad():
        out = proc(attn, hs, encoder_hidden_states=(text, [ip]))
        n_first = len(calls)
        out2 = proc(attn, hs * 0.5, encoder_hidden_states=(text, [ip]))          # the next denoising step: same tokens, new hidden states
    assert out.shape == hs.shape and out.dtype == dtype
    assert n_first == 6 and len(calls) - n_first == 2                             # q, k, v, k_ip, v_ip, out  |  q, out: K / V come from the cache
    tol = 1.5e-2 if dtype == torch.float16 else 6e-2
    np.testing.assert_allclose(out.float().cpu().numpy(), ref_out.float().cpu().numpy(), atol=tol)
    assert torch.isfinite(out2).all()
    with torch.no_grad():                                                          # 4-D (B,C,H,W) input path of the diffusers processors
        out4 = proc(attn, hs.transpose(1, 2).reshape(B, dim, 16, 16), encoder_hidden_states=(text, [ip]))
    np.testing.assert_allclose(out4.float().cpu().numpy(), out.transpose(1, 2).reshape(B, dim, 16, 16).float().cpu().numpy(), atol=1e-3)


def test_processor_kv_cache_never_serves_another_tensors_projections():
    check_processor_kv_cache_never_serves_another_tensors_projections("cuda", dim=640, heads=10, cross=2048, HW=256)


@pytest.mark.parametrize("sched,steps,guidance", [("ddim", 4, 5.0), ("euler", 3, 0.0), ("euler", 1, 0.0), ("ddim", 3, 0.0)])
def test_sampling_loop_matches_the_oracle(sched, steps, guidance):
    """F1: generate_ip_adapter_embeds on a small SDXL-shaped stand-in (one layer per stage, 16 x 16 latents), fp16.  The oracle rounds to fp16 where
    the product stores a 16-bit tensor; what is left is accumulation order and the kernels' 16-bit probabilities (budget: latents within 1e-2)."""
    pipe, W, cfg = small_pipe(sched, "cuda")
    emb = torch.randn(2, 1024, generator=torch.Generator().manual_seed(3)).cuda().half()
    out = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=steps, guidance_scale=guidance,
                                          generator=torch.Generator().manual_seed(11)).images
    ref = oracle_loop(pipe, W, cfg, sched, steps, guidance, emb.float().cpu().numpy().astype(np.float64), seed=11)
    assert out.shape == (2, 4, 16, 16) and out.dtype == torch.float16
    d = np.abs(out.float().cpu().numpy() - ref)
    assert d.max() < 1e-2 * max(1.0, np.abs(ref).max()), (d.max(), np.abs(ref).max())


def test_img2img_start_from_a_low_level_latent():
    """f3 (custom_pipeline_low_level.py:331-389): strength 0.5 of 4 steps skips the first two timesteps; start = latent * 0.13025 + unit noise"""
    pipe, W, cfg = small_pipe("ddim", "cuda")
    emb = torch.randn(1, 1024, generator=torch.Generator().manual_seed(4)).cuda().half()
    low = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    out = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=4, guidance_scale=0.0, img2img_strength=0.5,
                                          low_level_latent=low, generator=torch.Generator().manual_seed(12)).images
    ref = oracle_loop(pipe, W, cfg, "ddim", 4, 0.0, emb.float().cpu().numpy().astype(np.float64), seed=12, low_level_latent=low.numpy().astype(np.float64),
                      strength=0.5)
    assert np.abs(out.float().cpu().numpy() - ref).max() < 1e-2 * max(1.0, np.abs(ref).max())


def test_generator4embeds_runs_the_loop_on_the_stand_in_and_batches_are_independent():
    from eeg_image_decode_amd.sdxl import Generator4Embeds
    pipe, _, _ = small_pipe("ddim", "cuda")
    g = Generator4Embeds(num_inference_steps=3, device="cuda", pipe=pipe)
    embs = torch.randn(3, 1024, generator=torch.Generator().manual_seed(6))
    lat = torch.randn(3, 4, 16, 16, generator=torch.Generator().manual_seed(7)).cuda().half()
    batch = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=embs.cuda().half(), num_inference_steps=3, guidance_scale=0.0, latents=lat).images
    for i in range(3):                                           # 8 images per GPU are 8 independent chains
        one = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=embs[i:i + 1].cuda().half(), num_inference_steps=3, guidance_scale=0.0,
                                              latents=lat[i:i + 1]).images
        np.testing.assert_allclose(one.float().cpu().numpy(), batch[i:i + 1].float().cpu().numpy(), atol=2e-3)
    img = g.generate(embs[0], generator=torch.Generator().manual_seed(1))
    assert img.shape == (4, 16, 16) and torch.isfinite(img).all()


def test_generator4embeds_fails_loudly_without_diffusers_or_a_pipeline():
    from eeg_image_decode_amd._lib import EegclipError
    from eeg_image_decode_amd.sdxl import Generator4Embeds
    try:
        import diffusers  # noqa: F401
        pytest.skip("diffusers present")
    except ImportError:
        with pytest.raises(EegclipError):
            Generator4Embeds(4)


def test_full_size_stand_in_one_step_is_finite():
    """the bench configuration's shapes (1024 px: 4096 / 1024 tokens, all 70 attention positions, 2 images x CFG) for one DDIM step"""
    from eeg_image_decode_amd.sdxl import DDIMScheduler, SDXLShapedUNet, StandInSDXLPipeline
    pipe = StandInSDXLPipeline(SDXLShapedUNet(), DDIMScheduler(), device="cuda", default_sample_size=128)
    emb = torch.randn(2, 1024, device="cuda", dtype=torch.float16)
    out = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=1, guidance_scale=5.0,
                                          generator=torch.Generator(device="cuda").manual_seed(0)).images
    assert out.shape == (2, 4, 128, 128) and torch.isfinite(out).all() and float(out.float().std()) > 0.1


def test_full_size_stand_in_one_step_matches_the_oracle():
    """F1 at the bench width: 128 x 128 latents (4096 / 1024 tokens), all 70 cross-attention positions, 2 images x the CFG pair,
    one DDIM step -- EVERY latent of the output against oracle/sdxl_pipeline.py (numpy float64 with fp16 rounding at the product's storage
    points; ~30 s of host time), not only isfinite"""
    from eeg_image_decode_amd.sdxl import DDIMScheduler, SDXLShapedUNet, StandInSDXLPipeline
    unet = SDXLShapedUNet()
    pipe = StandInSDXLPipeline(unet, DDIMScheduler(), device="cuda", default_sample_size=128)
    W = {k: v.detach().float().cpu().numpy().astype(np.float64) for k, v in unet.state_dict().items()}
    emb = torch.randn(2, 1024, generator=torch.Generator().manual_seed(4)).half()
    out = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb.cuda(), num_inference_steps=1, guidance_scale=5.0,
                                          generator=torch.Generator().manual_seed(12)).images
    ref = oracle_loop(pipe, W, ((4, 20, 10, 30, 6), 1.0), "ddim", 1, 5.0, emb.float().numpy().astype(np.float64), seed=12)
    assert out.shape == (2, 4, 128, 128)
    d = np.abs(out.float().cpu().numpy() - ref)
    assert d.max() < 1e-2 * max(1.0, np.abs(ref).max()), (d.max(), np.abs(ref).max())
    assert d.mean() < 1e-3, d.mean()
