#!/usr/bin/env python3
"""Golden fixtures for the joint-subject ATMS: runs THE REFERENCE's
Retrieval/ATMS_retrieval_joint_train.py:ATMS(joint_train=True) (imported in place, third-party modules stubbed as in make_golden.py;
its dataset module needs open_clip and is stubbed too) on synthetic inputs and stores outputs only:
    tests/golden/joint_keys.json   state_dict keys / shapes (87 entries: one value-embedding Linear per subject)
    tests/golden/joint.npz         eval embeddings for uniform and mixed subject ids; train-mode (dropout p = 0) loss and gradients for a
                                   batch of mixed subjects, including which value-embedding layers receive no gradient

    python tests/golden/make_golden_joint.py
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import atms as oatms  # noqa: E402  (key/shape/kind spec only)

N_SUBJ = 10


def main():
    mg.import_reference()
    mg._stub("eegdatasets_joint_subjects", EEGDataset=object)
    spec = importlib.util.spec_from_file_location("ref_joint", os.path.join(mg.REF, "Retrieval", "ATMS_retrieval_joint_train.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    m = ref.ATMS(joint_train=True)
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(HERE, "joint_keys.json"), "w") as f:
        json.dump({"keys": keys, "n_params": sum(p.numel() for p in m.parameters())}, f, indent=0)
    ours = {k: list(s) for k, s, _ in oatms.state_spec(True, N_SUBJ)}
    assert ours == keys and list(ours) == list(keys), "oracle.state_spec(joint) drifted from the reference state_dict"
    state = mg.syn.make_state(mg.SEED + 30, oatms.state_spec(True, N_SUBJ))
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()}, strict=True)
    out = {}
    x = mg.t(mg.syn.eeg_batch(mg.SEED + 31, 8))
    ids_mixed = torch.tensor([0, 3, 3, 9, 1, 0, 7, 3])
    m.eval()
    with torch.no_grad():
        out["emb_uniform4"] = m(x, torch.full((8,), 4, dtype=torch.long)).numpy()
        out["emb_mixed"] = m(x, ids_mixed).numpy()
    out["ids_mixed"] = ids_mixed.numpy()
    # train mode, dropout p = 0, mixed subjects: loss (0.99 / 0.01 mix, ATMS_retrieval_joint_train.py:228-233) and gradients
    mg.zero_dropout(m)
    m.train()
    B = 12
    xb = mg.t(mg.syn.eeg_batch(mg.SEED + 32, B))
    img, txt = mg.t(mg.syn.unit_features(mg.SEED + 32, B, tag="img")), mg.t(mg.syn.unit_features(mg.SEED + 32, B, tag="txt"))
    ids_b = torch.tensor([2, 2, 5, 0, 5, 5, 2, 0, 9, 2, 5, 0])
    z = m(xb, ids_b).float()
    s = m.logit_scale
    loss = 0.99 * m.loss_func(z, img, s) + 0.01 * m.loss_func(z, txt, s)
    loss.backward()
    out["train_ids"] = ids_b.numpy()
    out["train_loss"] = np.float64(loss.item())
    out["train_z"] = z.detach().numpy()
    none_keys = []
    for k, p in m.named_parameters():
        if p.grad is None:
            none_keys.append(k)
        else:
            g = p.grad.detach()
            out["gnorm:" + k] = np.float32(g.norm().item())
            if "value_embedding" in k or k == "proj_eeg.0.bias":
                out["grad:" + k] = g.numpy().reshape(-1)[:512].copy()
    out["none_grad_keys"] = np.array(none_keys)
    np.savez_compressed(os.path.join(HERE, "joint.npz"), **out)
    print("wrote joint_keys.json, joint.npz; loss", loss.item(), "no-grad keys:", len(none_keys))


if __name__ == "__main__":
    main()
