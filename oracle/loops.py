"""Oracle: the contrastive train / eval loops (torch-CPU restatement).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Pinned by tests/golden/train_loop.npz and
tests/golden/eval.npz.  Also the ``cpu_baseline`` ("port") leg of bench.py.

Reference lines restated (under /root/reference/Retrieval/ATMS_retrieval.py):
  :199-254  train_model   (fwd, img+text ClipLoss mix 0.99/0.01, bwd, AdamW, running accuracy)
  :258-362  evaluate_model (bs=1, k-way candidates = random.sample(others,k-1)+[label], argmax/top-5)
and the reconstruction-objective variant of the same loop, Generation/ATMS_reconstruction.py:191-249 (pinned by
tests/golden/recon_loop.npz).
"""
import random

import numpy as np
import torch

from . import atms as oatms
from . import loss as oloss

# parameters that receive no gradient in the reference
_DEAD_PREFIXES = ("subject_wise_linear.", "encoder.enc_embedding.mask_token",
                  "encoder.enc_embedding.temporal_embedding.",
                  "encoder.enc_embedding.subject_embedding.mask_embedding")
# live parameters whose gradient is identically zero in exact arithmetic, so the reference's own value is
# round-off noise (and Adam turns that noise into +-lr steps): a bias added to every key is cancelled by
# softmax shift-invariance; a conv bias in front of a train-mode BatchNorm is cancelled by the mean subtraction.
ZERO_GRAD_KEYS = ("encoder.encoder.attn_layers.0.attention.key_projection.bias",
                  "enc_eeg.0.tsconv.0.bias", "enc_eeg.0.tsconv.4.bias")
_BUFFERS = ("position_embedding.pe", "running_mean", "running_var", "num_batches_tracked")


def is_buffer(key):
    return any(key.endswith(b) for b in _BUFFERS)


def is_dead(key):
    return key.startswith(_DEAD_PREFIXES)


def torch_state(state_np, dtype=torch.float32):
    out = {}
    for k, v in state_np.items():
        tns = torch.from_numpy(np.array(v))
        out[k] = tns.to(dtype) if tns.is_floating_point() else tns
    return out


class OracleTrainer:
    """Holds a state dict + AdamW moments; ``step`` = one iteration of the reference batch loop."""

    def __init__(self, state, lr=3e-4, p_scale=1.0, objective="retrieval"):
        """objective: "retrieval" = 0.99/0.01 image/text InfoNCE mix (Retrieval/ATMS_retrieval.py:224-229);
        "reconstruction" = 10 * (0.9 MSE + 0.1 image InfoNCE) (Generation/ATMS_reconstruction.py:222-228)"""
        self.objective = objective
        self.P = {k: v.clone() for k, v in state.items()}
        self.lr = lr
        self.p_scale = p_scale
        self.t = 0
        self.m, self.v = {}, {}
        self.params = [k for k in self.P if not is_buffer(k)]

    def loss_and_grads(self, x, subject_ids, img, txt, train=True, masks=None):
        live = {}
        for k in self.params:
            self.P[k] = self.P[k].detach().requires_grad_(True)
        want = {}
        z = oatms.atms_forward(self.P, x, subject_ids, train=train, masks=masks, p_scale=self.p_scale, want=want)
        s = self.P["logit_scale"]
        loss = oloss.mixed_loss(z, img, txt, s) if self.objective == "retrieval" else oloss.reconstruction_loss(z, img, s)
        grads = torch.autograd.grad(loss, [self.P[k] for k in self.params], allow_unused=True)
        for k, g in zip(self.params, grads):
            live[k] = g
        for k in self.params:
            self.P[k] = self.P[k].detach()
        return loss.detach(), z.detach(), live, want

    def step(self, x, subject_ids, img, txt, masks=None):
        loss, z, grads, want = self.loss_and_grads(x, subject_ids, img, txt, True, masks)
        # BatchNorm running statistics (train mode side effect)
        n1 = x.shape[0] * 63 * 36
        n2 = x.shape[0] * 36
        for tag, n, mk, vk in (("enc_eeg.0.tsconv.2.", n1, "bn1_mean", "bn1_var"), ("enc_eeg.0.tsconv.5.", n2, "bn2_mean", "bn2_var")):
            rm, rv = oatms.bn_running_update(self.P[tag + "running_mean"], self.P[tag + "running_var"],
                                             want[mk].detach(), want[vk].detach(), n)
            self.P[tag + "running_mean"], self.P[tag + "running_var"] = rm, rv
            self.P[tag + "num_batches_tracked"] = self.P[tag + "num_batches_tracked"] + 1
        self.t += 1
        for k in self.params:
            g = grads[k]
            if g is None:          # torch optimizers skip params whose grad is None entirely
                continue
            if k not in self.m:
                self.m[k] = np.zeros(g.shape, np.float32)
                self.v[k] = np.zeros(g.shape, np.float32)
            p = self.P[k].numpy().copy()
            oloss.adamw_step(p, g.numpy().astype(np.float32), self.m[k], self.v[k], self.t, lr=self.lr)
            self.P[k] = torch.from_numpy(p)
        return loss, z


def train_epoch(trainer, sub_id, batches, img_features_all):
    """train_model (:199-254) on an iterable of (eeg, labels, text, text_feat, img, img_feat)."""
    class_feats = img_features_all[::10]
    total, correct, n = 0.0, 0, 0
    feats = []
    for (x, labels, _t, txt, _p, img) in batches:
        ids = torch.full((x.shape[0],), sub_id, dtype=torch.long)
        loss, z = trainer.step(x, ids, img, txt)
        total += float(loss)
        # NOTE the reference scores with the POST-step logit_scale but PRE-step features (:240-241)
        pred = oloss.train_accuracy_predictions(z, class_feats, trainer.P["logit_scale"])
        correct += int((pred == labels).sum())
        n += x.shape[0]
        feats.append(z)
    return total / len(batches), correct / n, torch.cat(feats, 0)


def evaluate(P, sub_id, samples, img_features_all, txt_features_all, k, rng=random):
    """evaluate_model (:258-362).  ``samples``: list of (x (1,63,250), label int, txt (1,1024), img (1,1024)).
    Consumes ``rng.sample`` exactly like the reference (one draw per sample, a second one for k<200)."""
    total, correct, top5c, n = 0.0, 0, 0, 0
    all_labels = set(range(txt_features_all.shape[0]))
    s = P["logit_scale"]
    for (x, label, txt, img) in samples:
        ids = torch.full((x.shape[0],), sub_id, dtype=torch.long)
        z = oatms.atms_forward(P, x, ids, train=False)
        total += float(oloss.mixed_loss(z, img, txt, s))
        possible = list(all_labels - {label})
        selected = rng.sample(possible, k - 1) + [label]
        cand = img_features_all[selected]
        if k in (50, 100, 2, 4, 10):
            selected = rng.sample(possible, k - 1) + [label]      # re-sampled AFTER the features were gathered (:328)
        top1, top = oloss.kway_retrieval(z[0], cand, s)
        if selected[top1] == label:
            correct += 1
        if k in (200, 100, 50) and label in [selected[i] for i in top]:
            top5c += 1
        n += 1
    return total / len(samples), correct / n, top5c / n
