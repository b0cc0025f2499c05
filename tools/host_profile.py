"""where the HOST time of a training step goes: cProfile over N un-synchronised steps (the GPU queue absorbs them) + enqueue time per step
with the C plan executor and with the Python one"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eeg_image_decode_amd import retrieval


def run(n, step):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


def main():
    model, opt, pool, classes = bench.build(1, 0, 256)
    loss_acc = []
    correct = torch.zeros(1, dtype=torch.int32, device="cuda")

    def step(i):
        d = pool[i % len(pool)]
        retrieval.contrastive_step(model, opt, d["eeg"], 1, d["img"], d["txt"], d["labels"], classes, loss_acc, correct)
    for i in range(20):
        step(i)
    for mode in ("c", "python", "c"):
        for pl in model._engine().plans.values():
            pl.use_c_executor = mode == "c"
        run(20, step)
        enq, tot = run(100, step)
        print(f"executor={mode:6s}  host enqueue {enq:.3f} ms/step   wall {tot:.3f} ms/step", flush=True)
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for i in range(200):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
