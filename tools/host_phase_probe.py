"""host time of the pieces of one training step (perf_counter around un-synchronised calls; diagnosis)"""
import os, sys, time, timeit
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eeg_image_decode_amd import retrieval, _abi

model, opt, pool, classes = bench.build(1, 0, 256)
correct = torch.zeros(1, dtype=torch.int32, device="cuda")
acc = {}


def tick(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t


def step(i):
    d = pool[i % len(pool)]
    t = time.perf_counter()
    opt.zero_grad()
    t = tick("zero_grad", t)
    ids = retrieval._uniform_ids(256, 1, d["eeg"].device)
    z = model(d["eeg"], ids).float()
    t = tick("encoder forward", t)
    lf = model.loss_func
    lf._unit_upstream_grad = True
    loss = lf.forward_mixed(z, [(d["img"], 0.99), (d["txt"], 0.01)], model.logit_scale)
    lf._unit_upstream_grad = False
    t = tick("loss forward", t)
    loss.backward()
    t = tick("backward (autograd: loss + encoder)", t)
    retrieval._accumulate_accuracy(z, classes, model.logit_scale, d["labels"], 256, correct)
    t = tick("accuracy readout", t)
    opt.step(zero_grad=True)
    t = tick("optimizer", t)


for i in range(50):
    step(i)
acc.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 300
for i in range(N):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host {1e3 * (t1 - t0) / N:.3f} ms/step, wall {1e3 * (time.perf_counter() - t0) / N:.3f} ms/step")
for k, v in acc.items():
    print(f"  {k:40s} {1e6 * v / N:8.1f} us")
print("dim() alone: %.2f us" % (1e6 * min(timeit.repeat(lambda: _abi.dim(5), number=1000, repeat=5)) / 1000))
print("raw_stream() alone: %.2f us" % (1e6 * min(timeit.repeat(lambda: retrieval.raw_stream(), number=1000, repeat=5)) / 1000))
x = torch.empty(4, device="cuda")
print("data_ptr() alone: %.2f us" % (1e6 * min(timeit.repeat(lambda: x.data_ptr(), number=1000, repeat=5)) / 1000))
print("torch.empty alone: %.2f us" % (1e6 * min(timeit.repeat(lambda: torch.empty(256, 1024, device='cuda'), number=1000, repeat=5)) / 1000))
