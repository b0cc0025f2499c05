"""Oracle: CLIP-symmetric InfoNCE, retrieval metrics and AdamW (torch-CPU / numpy restatement).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Pinned by tests/golden/loss_*.npz,
tests/golden/dist_*.npz, tests/golden/train_*.npz.

Reference lines restated (under /root/reference):
  models/loss.py:100-141                   ClipLoss.forward (raw logit_scale multiplier, mean CE, /2)
  models/loss.py:20-75                     gather_features (3 gather modes)
  Retrieval/ATMS_retrieval.py:229-234      0.99*img + 0.01*text mix
  Retrieval/ATMS_retrieval.py:241-250      train accuracy vs img_features_all[::10]
  Retrieval/ATMS_retrieval.py:297-355      k-way retrieval (label last, argmax / top-5)
  Retrieval/ATMS_retrieval.py:548          AdamW(lr) with torch defaults
"""
import numpy as np
import torch


def clip_loss(a, b, logit_scale):
    """L = 1/2 [ CE(s A B^T, arange) + CE(s B A^T, arange) ], mean reduction (loss.py:122-140).
    ``logit_scale`` is used RAW (no exp) -- SURVEY.md section 9 quirk 1."""
    s = logit_scale * a @ b.T
    st = logit_scale * b @ a.T
    n = a.shape[0]
    idx = torch.arange(n)
    row = torch.logsumexp(s, dim=1) - s[idx, idx]
    col = torch.logsumexp(st, dim=1) - st[idx, idx]
    return 0.5 * (row.mean() + col.mean())


def clip_loss_grads(a, b, logit_scale):
    """Closed-form gradients: G = (P_row + P_col - 2I)/(2N);
    dA = s G B ; dB = s G^T A ; ds = sum(G * A B^T).  float64 inside."""
    a64, b64 = a.double(), b.double()
    s = float(logit_scale)
    raw = a64 @ b64.T
    S = s * raw
    n = a.shape[0]
    G = (torch.softmax(S, dim=1) + torch.softmax(S, dim=0) - 2 * torch.eye(n, dtype=torch.float64)) / (2 * n)
    return (s * G @ b64), (s * G.T @ a64), (G * raw).sum()


def clip_loss_local(a_loc, b_loc, a_all, b_all, logit_scale, rank):
    """Row-sharded form, local_loss=True (loss.py:113-115,129-130): rank r scores its n rows against
    all N gathered columns, labels offset by n*r."""
    n = a_loc.shape[0]
    lab = torch.arange(n) + n * rank
    s1 = logit_scale * a_loc @ b_all.T
    s2 = logit_scale * b_loc @ a_all.T
    ar = torch.arange(n)
    r = torch.logsumexp(s1, 1) - s1[ar, lab]
    c = torch.logsumexp(s2, 1) - s2[ar, lab]
    return 0.5 * (r.mean() + c.mean())


def mixed_loss(z, img, txt, logit_scale, alpha=0.99):
    """ATMS_retrieval.py:229-234."""
    return alpha * clip_loss(z, img, logit_scale) + (1 - alpha) * clip_loss(z, txt, logit_scale)


def reconstruction_loss(z, img, logit_scale, alpha=0.90):
    """Generation/ATMS_reconstruction.py:222-228: 10 * (alpha * MSE(z, img) + (1 - alpha) * ClipLoss(z, img)); the text loss is computed
    by the reference but does not enter the objective."""
    return alpha * torch.nn.functional.mse_loss(z, img) * 10 + (1 - alpha) * clip_loss(z, img, logit_scale) * 10


def train_accuracy_predictions(z, class_feats, logit_scale):
    """argmax over logit_scale * z @ class_feats.T (ATMS_retrieval.py:241-246); ties -> lowest index."""
    return torch.argmax(logit_scale * z @ class_feats.T, dim=1)


def kway_retrieval(z_i, cand_feats, logit_scale, topk=5):
    """One query against k candidates (ATMS_retrieval.py:305-320): returns (argmax, top-k indices)."""
    logits = logit_scale * z_i @ cand_feats.T
    top1 = int(torch.argmax(logits))
    k = min(topk, logits.numel())
    return top1, torch.topk(logits, k, largest=True).indices.tolist()


def adamw_step(p, g, m, v, step, lr=3e-4, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor math, torch defaults.  numpy, in place.
    ``step`` is the 1-based step count AFTER this update."""
    p *= (1.0 - lr * wd)
    m *= b1
    m += (1 - b1) * g
    v *= b2
    v += (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p -= (lr / bc1) * (m / denom)
    return p, m, v


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay) -- the prior's optimizer (diffusion_prior.py:286)."""
    return adamw_step(p, g, m, v, step, lr, b1, b2, eps, wd=0.0)
