"""per-step HOST time of the training loop over many steps: prints the outliers (a one-time 50-150 ms stall around the 50th step of a process was seen in
round 4) -- python tools/host_stalls.py [steps] [--no-settle]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eeg_image_decode_amd import retrieval

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
if "--no-settle" in sys.argv:                      # the product freezes the heap before its first step (retrieval.settle_gc): show the stall it avoids
    retrieval._GC_SETTLED = True
model, opt, pool, classes = bench.build(1, 0, 256)
loss_acc, correct = [], torch.zeros(1, dtype=torch.int32, device="cuda")
ts = []
gc_events = []
gc.callbacks.append(lambda phase, info: gc_events.append((len(ts), phase, info.get("generation"), time.perf_counter())))
for i in range(n):
    d = pool[i % len(pool)]
    a = time.perf_counter()
    retrieval.contrastive_step(model, opt, d["eeg"], 1, d["img"], d["txt"], d["labels"], classes, loss_acc, correct)
    ts.append((time.perf_counter() - a) * 1e3)
torch.cuda.synchronize()
import statistics
print("median host ms/step %.3f" % statistics.median(ts))
for i, t in enumerate(ts):
    if t > 3.0:
        print("step", i, "host %.1f ms" % t)
g2 = [(s, ph, gen) for s, ph, gen, _ in gc_events if gen == 2 and ph == "start"]
print("gen-2 collections at steps:", [s for s, _, _ in g2][:20])
st = {}
for j in range(len(gc_events) - 1):
    s, ph, gen, t0 = gc_events[j]
    if ph == "start" and gc_events[j + 1][1] == "stop":
        dt = (gc_events[j + 1][3] - t0) * 1e3
        if dt > 2.0:
            print("gc gen", gen, "at step", s, "took %.1f ms" % dt)
