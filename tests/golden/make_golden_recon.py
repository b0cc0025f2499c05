#!/usr/bin/env python3
"""Golden fixture for the reconstruction-objective train loop: runs THE REFERENCE's
Generation/ATMS_reconstruction.py:train_model (imported in place, third-party modules stubbed exactly as in make_golden.py) on the
same synthetic 3-batch loader as train_loop.npz and stores outputs only -> tests/golden/recon_loop.npz.

    python tests/golden/make_golden_recon.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.import_reference()                                   # installs the stubs and the reference's sys.path entries
    sys.path.insert(0, os.path.join(mg.REF, "Generation"))
    spec = importlib.util.spec_from_file_location("ref_recon", os.path.join(mg.REF, "Generation", "ATMS_reconstruction.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    n_classes, B = 20, 16
    img_all = mg.t(mg.syn.unit_features(mg.SEED + 4, n_classes * 10, tag="imgall"))
    txt_all = mg.t(mg.syn.unit_features(mg.SEED + 4, n_classes, tag="txtall"))
    m = ref.ATMS()
    mg.load_synth(m)
    mg.zero_dropout(m)
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4)
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    out, losses, accs = {}, [], []
    for ep in range(2):
        batches = mg._make_batches(mg.SEED + 4, 3, B, n_classes, img_all, txt_all)
        l, a, feats = ref.train_model("sub-01", m, mg._ListLoader(batches), opt, "cpu", txt_all, img_all, None)
        losses.append(l)
        accs.append(a)
        if ep == 0:
            out["feats_ep0"] = feats.detach().numpy()[:, :64].copy()
    out["losses"] = np.asarray(losses, np.float64)
    out["accs"] = np.asarray(accs, np.float64)
    for k, p in m.named_parameters():
        out["dnorm:" + k] = np.float32((p.detach() - before[k]).flatten().norm().item())
        out["pnorm:" + k] = np.float32(p.detach().norm().item())
    np.savez_compressed(os.path.join(HERE, "recon_loop.npz"), **out)
    print("wrote recon_loop.npz", losses, accs)


if __name__ == "__main__":
    main()
