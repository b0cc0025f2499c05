"""Retrieval-accuracy parity on held-out synthetic pairs (north_star: top-1 / top-5 within +-0.1 % of the reference path).

The MI355X product (HIP kernels) and the CPU oracle (the pinned restatement of the reference: the reference itself cannot travel to the GPU
box) start from the same weights and see the same batches of class-structured EEG / target pairs (eeg_image_decode_amd.synthetic.learnable_pairs:
1000 training classes x 2 trials, dropout off so both runs are deterministic), 150 AdamW steps at batch 128 with the reference's 0.99 / 0.01
image / text InfoNCE mix.  Both are then scored exactly like the reference's test protocol -- 200 held-out classes, one averaged trial each,
evaluate_model / its oracle counterpart with the same seeded candidate lists for k = 200, 100, 50, 10, 4, 2 (Retrieval/ATMS_retrieval.py:258-362).
Prints one JSON object (profiles/r1_accuracy_parity.json: exact fp32 products, 200 held-out classes; profiles/r3_accuracy_parity.json: the default
split-bf16 GEMM arithmetic, 1000 held-out classes so that 0.1 % is one query).  The oracle half takes ~2 minutes of host CPU time."""
import json, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eeg_image_decode_amd import optim, retrieval, synthetic as syn
from eeg_image_decode_amd.atms import ATMS
from oracle import atms as oatms, loops as oloops            # the checker (tools/ = measurement harness, like bench.py's cpu_baseline leg)


def main(steps=150, B=128, n_train=1000, per=2, n_test=200, noise=0.5, seed=5):
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    eeg, lab, protos = syn.learnable_pairs(seed, n_train + n_test, per, noise=noise)
    tr = lab < n_train
    xtr, ltr = torch.from_numpy(eeg[tr]), torch.from_numpy(lab[tr])
    xte = torch.from_numpy(eeg[~tr].reshape(n_test, per, 63, 250).mean(1))            # averaged test trials, like the reference's test split
    p_tr, p_te = torch.from_numpy(protos[:n_train]), torch.from_numpy(protos[n_train:])
    state = syn.make_state(1, oatms.state_spec())
    order = np.random.default_rng(0)
    batches = [order.permutation(len(xtr))[:B] for _ in range(steps)]

    # ---- product on the GPU
    m = ATMS()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m = m.cuda().train()
    opt = optim.AdamW(m.parameters(), lr=3e-4)
    loss_acc, correct = torch.zeros((), device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    xg, pg, lg = xtr.cuda(), p_tr.cuda(), ltr.cuda()
    t0 = time.perf_counter()
    for idx in batches:
        i = torch.from_numpy(idx).cuda()
        tgt = pg[lg[i]]
        retrieval.contrastive_step(m, opt, xg[i].contiguous(), 1, tgt, tgt, lg[i], pg, loss_acc, correct)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0

    # ---- oracle on the host CPU
    T = oloops.OracleTrainer(oloops.torch_state(state), p_scale=0.0)
    t0 = time.perf_counter()
    last = None
    for idx in batches:
        tgt = p_tr[ltr[idx]]
        last, _ = T.step(xtr[idx], torch.full((B,), 1).long(), tgt, tgt)
    t_cpu = time.perf_counter() - t0

    # ---- the reference's test protocol on both
    test_items = [(xte[i:i + 1], torch.tensor([i]), "", p_te[i:i + 1], "", p_te[i:i + 1]) for i in range(n_test)]
    out = {"gemm_arithmetic": os.environ.get("EEGCLIP_GEMM_PRECISION", "bf16x3"), "steps": steps, "batch": B, "train_classes": n_train, "test_classes": n_test, "train_seconds_gpu": round(t_gpu, 2),
           "train_seconds_cpu_oracle": round(t_cpu, 1), "mean_train_loss_gpu": round(float(loss_acc) / steps, 4), "final_step_loss_oracle": round(float(last), 4)}
    with torch.no_grad():
        zg = m.eval()(xte.cuda(), 1).cpu()
    zo = oatms.atms_forward(T.P, xte, torch.full((n_test,), 1).long(), train=False)
    out["test_embedding_max_abs_diff"] = float((zg - zo).abs().max())
    out["test_embedding_min_cosine"] = float(torch.nn.functional.cosine_similarity(zg, zo).min())
    top_g, top_o = (zg @ p_te.T).topk(5, 1).indices, (zo @ p_te.T).topk(5, 1).indices
    out["top1_index_mismatches"] = int((top_g[:, 0] != top_o[:, 0]).sum())
    out["top5_list_mismatches"] = int((top_g != top_o).any(1).sum())
    res = {}
    for k in (200, 100, 50, 10, 4, 2):
        random.seed(1234 + k)
        _, acc_g, top5_g = retrieval.evaluate_model("sub-01", m, test_items, "cuda", p_te, p_te, k, None)
        random.seed(1234 + k)
        samples = [(x, int(l), tf, imf) for (x, l, _, tf, _, imf) in test_items]
        _, acc_o, top5_o = oloops.evaluate(T.P, 1, samples, p_te, p_te, k)
        res[f"k{k}"] = {"top1_gpu": acc_g, "top1_oracle": acc_o, "top5_gpu": top5_g, "top5_oracle": top5_o,
                        "abs_diff_top1": abs(acc_g - acc_o), "abs_diff_top5": abs(top5_g - top5_o)}
    out["k_way"] = res
    out["max_abs_accuracy_diff"] = max(max(v["abs_diff_top1"], v["abs_diff_top5"]) for v in res.values())
    print(json.dumps(out))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:4]]            # steps, batch, held-out classes
    main(*a[:2], **({"n_test": a[2]} if len(a) > 2 else {}))
