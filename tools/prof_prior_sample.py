"""rocprofv3 target: diffusion-prior sampling only (8 embeddings, 50 DDPM steps, CFG), launch by launch so every kernel shows up by name"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EEGCLIP_PRIOR_GRAPH", "0")
import torch
from eeg_image_decode_amd.prior import DiffusionPriorUNet, Pipe

pipe = Pipe(DiffusionPriorUNet(cond_dim=1024, dropout=0.1), device="cuda")
c = torch.randn(8, 1024, device="cuda")
gen = torch.Generator(device="cuda").manual_seed(1)
for _ in range(3):
    pipe.generate(c_embeds=c, num_inference_steps=50, guidance_scale=5.0, generator=gen)
torch.cuda.synchronize()
