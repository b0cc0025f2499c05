"""host enqueue time per training step with the cyclic garbage collector on / frozen / off (diagnosis)"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eeg_image_decode_amd import retrieval

model, opt, pool, classes = bench.build(1, 0, 256)
loss_acc, correct = [], torch.zeros(1, dtype=torch.int32, device="cuda")


def step(i):
    d = pool[i % len(pool)]
    retrieval.contrastive_step(model, opt, d["eeg"], 1, d["img"], d["txt"], d["labels"], classes, loss_acc, correct)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3, (time.perf_counter() - t0) / n * 1e3


for i in range(50):
    step(i)
for mode in ("on", "freeze", "off", "on"):
    gc.enable(); gc.unfreeze()
    if mode == "freeze":
        gc.collect(); gc.freeze()
    if mode == "off":
        gc.collect(); gc.disable()
    run(50)
    c0 = gc.get_stats()
    enq, tot = run(300)
    c1 = gc.get_stats()
    print(f"gc {mode:6s}: host enqueue {enq:.3f} ms/step, wall {tot:.3f} ms/step, collections per generation during the run: {[b['collections'] - a['collections'] for a, b in zip(c0, c1)]}, tracked objects {len(gc.get_objects())}", flush=True)
