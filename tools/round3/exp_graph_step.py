"""experiment: replay the whole contrastive training step from one HIP graph (seeds and optimizer step count baked: timing only) to see how
much of the 1.59 ms step is host-induced idle time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

model, opt, pool, classes = bench.build(1, 0, 256)
from eeg_image_decode_amd import retrieval
loss_acc = torch.zeros((), device="cuda")
correct = torch.zeros(1, dtype=torch.int32, device="cuda")
d = pool[0]


def step():
    retrieval.contrastive_step(model, opt, d["eeg"], 1, d["img"], d["txt"], d["labels"], classes, loss_acc, correct)


for _ in range(300):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step()
torch.cuda.synchronize()
print(f"eager  {1e3 * (time.perf_counter() - t0) / 100:.4f} ms/step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    g.replay()
torch.cuda.synchronize()
print(f"graph  {1e3 * (time.perf_counter() - t0) / 100:.4f} ms/step")
