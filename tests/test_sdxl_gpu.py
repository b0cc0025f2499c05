"""Row F2 on the MI355X: the cross-attention processor (diffusers call signature) against the numpy oracle, on a duck-typed
diffusers `Attention` module with SDXL shapes (640 ch / 10 heads at 64x64 latents is too slow for the oracle; a 16x16 crop is used)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import sdxl_attn

pytestmark = pytest.mark.gpu


class FakeAttention(nn.Module):
    """the attributes of diffusers.models.attention_processor.Attention that a processor touches"""

    def __init__(self, dim, cross_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(cross_dim, dim, bias=False)
        self.to_v = nn.Linear(cross_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("dim,heads", [(640, 10), (1280, 20)])
def test_ip_adapter_cross_attention_processor(dtype, dim, heads):
    from eeg_image_decode_amd.sdxl import HIPIPAdapterAttnProcessor
    torch.manual_seed(0)
    B, HW, cross = 2, 256, 2048
    attn = FakeAttention(dim, cross, heads).cuda().to(dtype)
    proc = HIPIPAdapterAttnProcessor(dim, cross, scale=1.0).cuda().to(dtype)
    hs = torch.randn(B, HW, dim, device="cuda", dtype=dtype)
    text = torch.randn(B, 77, cross, device="cuda", dtype=dtype)
    ip = torch.randn(B, 1, 4, cross, device="cuda", dtype=dtype)
    with torch.no_grad():
        out = proc(attn, hs, encoder_hidden_states=(text, [ip]))
    assert out.shape == hs.shape and out.dtype == dtype
    # oracle on the same 16-bit projections
    with torch.no_grad():
        q, k, v = attn.to_q(hs), attn.to_k(text), attn.to_v(text)
        kip, vip = proc.to_k_ip(ip.flatten(1, 2)), proc.to_v_ip(ip.flatten(1, 2))
        ref = sdxl_attn.cross_attention(*(t.float().cpu().numpy() for t in (q, k, v)), heads, kip.float().cpu().numpy(), vip.float().cpu().numpy(), 1.0)
        ref_out = attn.to_out[0](torch.tensor(ref, dtype=torch.float32).cuda().to(dtype))
    tol = 1e-2 if dtype == torch.float16 else 4e-2
    np.testing.assert_allclose(out.float().cpu().numpy(), ref_out.float().cpu().numpy(), atol=tol)
    # 4-D (B,C,H,W) input path of the diffusers processors
    with torch.no_grad():
        out4 = proc(attn, hs.transpose(1, 2).reshape(B, dim, 16, 16), encoder_hidden_states=(text, [ip]))
    np.testing.assert_allclose(out4.float().cpu().numpy(), out.transpose(1, 2).reshape(B, dim, 16, 16).float().cpu().numpy(), atol=1e-3)


def test_generator4embeds_fails_loudly_without_diffusers():
    from eeg_image_decode_amd._lib import EegclipError
    from eeg_image_decode_amd.sdxl import Generator4Embeds
    try:
        import diffusers  # noqa: F401
        pytest.skip("diffusers present")
    except ImportError:
        with pytest.raises(EegclipError):
            Generator4Embeds(4)
