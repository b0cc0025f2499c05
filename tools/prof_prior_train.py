"""rocprofv3 target / host-vs-GPU probe: diffusion-prior training steps at batch 1024 (inputs resident in HBM)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd.prior import DiffusionPriorUNet, Pipe

B, batches = 1024, 20
g = torch.Generator().manual_seed(0)
c, h = torch.randn(B * batches, 1024, generator=g).cuda(), torch.randn(B * batches, 1024, generator=g).cuda()
pipe = Pipe(DiffusionPriorUNet(cond_dim=1024, dropout=0.1), device="cuda")
dl = [{"c_embedding": c[i:i + B], "h_embedding": h[i:i + B]} for i in range(0, B * batches, B)]
pipe.train(dl, num_epochs=1, learning_rate=1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipe.train(dl, num_epochs=2, learning_rate=1e-3)
t1 = time.perf_counter()           # train() ends with one loss read-back per epoch, so this includes the GPU drain of the last epoch
torch.cuda.synchronize()
print(f"wall per step {1e3 * (t1 - t0) / (2 * batches):.3f} ms")
