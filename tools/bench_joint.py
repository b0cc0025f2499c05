"""Joint-subject ATM-S train step on one MI355X: the reference loops' uniform-id batch (one value-embedding GEMM) against
batches that mix all 10 subjects (one GEMM per subject over the subject-ordered batch; unordered batches are gathered first), B = 256.
Prints one JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eeg_image_decode_amd import optim, retrieval, synthetic as syn
from eeg_image_decode_amd.retrieval_joint import ATMS


def main(B=256, steps=30, warm=8):
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.from_numpy(syn.eeg_batch(1, B)).to(dev)
    img, txt = torch.from_numpy(syn.unit_features(1, B, tag="img")).to(dev), torch.from_numpy(syn.unit_features(1, B, tag="txt")).to(dev)
    cls = torch.from_numpy(syn.unit_features(2, 1654, tag="img")).to(dev)
    labels = torch.arange(B, device=dev)
    rng = np.random.default_rng(0)
    mixed = rng.integers(0, 10, B).tolist()
    out = {}
    for name, ids in (("uniform", 3), ("mixed_ordered", sorted(mixed)), ("mixed_unordered", mixed)):
        m = ATMS(joint_train=True).to(dev).train()
        opt = optim.AdamW(m.parameters(), lr=3e-4)
        loss_acc, correct = torch.zeros((), device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        run = lambda: retrieval.contrastive_step(m, opt, x, ids, img, txt, labels, cls, loss_acc, correct)
        for _ in range(warm):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        out[name] = {"ms_per_step": round(ms, 4), "samples_per_s": round(B / ms * 1e3, 1), "mean_loss": round(float(loss_acc) / (steps + warm), 4)}
    print(json.dumps({"workload": f"joint-subject ATM-S contrastive train step, B={B}, 10 subjects", **out}))


if __name__ == "__main__":
    main()
