"""experiment: is the prior's training step host-bound?  host time to ENQUEUE an epoch (Pipe.train reads the loss back once per epoch) vs the
time until the GPU has finished it"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd.prior import DiffusionPriorUNet, Pipe

B, batches = 1024, 40
g = torch.Generator().manual_seed(0)
c, h = torch.randn(B * batches, 1024, generator=g).cuda(), torch.randn(B * batches, 1024, generator=g).cuda()
pipe = Pipe(DiffusionPriorUNet(cond_dim=1024, dropout=0.1), device="cuda")
dl = [{"c_embedding": c[i:i + B], "h_embedding": h[i:i + B]} for i in range(0, B * batches, B)]
pipe.train(dl, num_epochs=1, learning_rate=1e-3)
torch.cuda.synchronize()
# patch float() sync out of the measurement: time only the enqueue of one epoch by making the epoch-end read-back lazy
import eeg_image_decode_amd.prior as pp
t0 = time.perf_counter()
real_print = print
marks = []
class _T(torch.Tensor):
    pass
orig_float = torch.Tensor.__float__
def lazy_float(self):
    marks.append(time.perf_counter())          # host reached the end of the epoch's enqueue
    return orig_float(self)
torch.Tensor.__float__ = lazy_float
pipe.train(dl, num_epochs=1, learning_rate=1e-3)
torch.Tensor.__float__ = orig_float
t1 = time.perf_counter()
print(f"host enqueue of the epoch: {1e3 * (marks[0] - t0) / batches:.3f} ms/step; until the GPU finished: {1e3 * (t1 - t0) / batches:.3f} ms/step")
