"""Oracle: ATM-S EEG encoder forward (functional restatement, torch-CPU).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Pinned against the imported
reference by tests/golden/atms_*.npz (tests/test_oracle_golden.py).

The forward is written as one flat function over a ``state_dict``-style dict of
tensors (reference key names) so that the same
weights can be loaded into the reference model, this oracle and the HIP model.

Reference lines restated (all under /root/reference):
  models/subject_layers/Embed.py:8-26      sinusoid table
  models/subject_layers/Embed.py:109-121   subject token (shared token if id >= 10)
  models/subject_layers/Embed.py:141-162   value embedding + PE + prepend token + dropout
  models/subject_layers/SelfAttention_Family.py:56-75,194-213   4-head attention, d_k = 62
  models/subject_layers/Transformer_EncDec.py:39-51,61-80      post-LN encoder layer + final LN
  Retrieval/ATMS_retrieval.py:87-93        keep tokens 0..62
  Retrieval/ATMS_retrieval.py:97-125,140-146  tsconv + projection + flatten
  Retrieval/ATMS_retrieval.py:157-167      projection head
"""
import math

import torch
import torch.nn.functional as F

N_HEADS = 4
D_MODEL = 250
D_HEAD = D_MODEL // N_HEADS  # 62  (SelfAttention_Family.py:184-185)
P_DROP_ENC = 0.25            # Config.dropout, ATMS_retrieval.py:53
P_DROP_CONV = 0.5            # ATMS_retrieval.py:109
P_DROP_PROJ = 0.5            # ATMS_retrieval.py:157
BN_EPS = 1e-5
LN_EPS = 1e-5
BN_MOMENTUM = 0.1

# the seven dropout sites of the train-mode forward, in execution order
DROPOUT_SITES = ("embed", "attn", "attn_out", "ffn_act", "ffn_out", "conv", "proj")


def sinusoid_table(n_pos, d_model=D_MODEL, dtype=torch.float32):
    """pe[p, 2i] = sin(p * exp(-2i ln(1e4)/d)), pe[p, 2i+1] = cos(.)   (Embed.py:12-20)."""
    pos = torch.arange(n_pos, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(n_pos, d_model, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def _drop(x, p, train, masks, site):
    """Inverted dropout.  ``masks[site]`` (bool keep-mask, same shape) makes it deterministic."""
    if not train or p == 0.0:
        return x
    if masks is not None and site in masks:
        keep = masks[site].to(x.dtype)
        return x * keep / (1.0 - p)
    return F.dropout(x, p, True)


def _bn_train(y, w, b, dims, eps=BN_EPS):
    """BatchNorm in train mode: biased batch variance over ``dims``."""
    mean = y.mean(dim=dims, keepdim=True)
    var = y.var(dim=dims, unbiased=False, keepdim=True)
    shape = [1] * y.dim()
    shape[1] = -1
    return (y - mean) / torch.sqrt(var + eps) * w.view(shape) + b.view(shape), mean.flatten(), var.flatten()


def _bn_eval(y, w, b, rm, rv, eps=BN_EPS):
    shape = [1] * y.dim()
    shape[1] = -1
    return (y - rm.view(shape)) / torch.sqrt(rv.view(shape) + eps) * w.view(shape) + b.view(shape)


def subject_token(P, subject_ids, B, prefix="encoder.enc_embedding.subject_embedding."):
    """Embed.py:116-121: any id >= table size (10) -> the shared token for the WHOLE batch."""
    table = P[prefix + "subject_embedding.weight"]
    if subject_ids is None or bool((subject_ids >= table.shape[0]).any()):
        return P[prefix + "shared_embedding"].expand(B, 1, -1)
    return table[subject_ids].unsqueeze(1)


def atms_forward(P, x, subject_ids, train=False, masks=None, p_scale=1.0, want=None):
    """(B,63,250) -> (B,1024).

    P        dict of tensors keyed like the reference ``ATMS.state_dict()``.
    train    train-mode semantics: dropout active (p * p_scale), BatchNorm uses batch stats.
    masks    optional dict site -> bool keep mask (see DROPOUT_SITES) for reproducible dropout.
    want     optional dict that receives named intermediates (for per-kernel tests) and the
             BatchNorm batch statistics (for the running-stat update).
    """
    B = x.shape[0]
    dt = x.dtype
    pre = "encoder.enc_embedding."
    lay = "encoder.encoder.attn_layers.0."
    keep = want if want is not None else {}
    p_enc, p_conv, p_proj = P_DROP_ENC * p_scale, P_DROP_CONV * p_scale, P_DROP_PROJ * p_scale

    # --- A1: DataEmbedding (Embed.py:141-162): each EEG channel is one token of width 250
    if pre + "value_embedding.weight" in P:
        h = F.linear(x, P[pre + "value_embedding.weight"], P[pre + "value_embedding.bias"])
    else:
        # joint-subject model (Embed.py:127-131,142-144; ATMS_retrieval_joint_train.py:172-192): one Linear per subject, chosen per sample
        h = torch.stack([F.linear(x[i], P[pre + f"value_embedding.{int(s)}.weight"], P[pre + f"value_embedding.{int(s)}.bias"])
                         for i, s in enumerate(subject_ids)])
    h = h + P[pre + "position_embedding.pe"][0, : x.shape[1]].to(dt)
    h = torch.cat([subject_token(P, subject_ids, B).to(dt), h], dim=1)       # (B,64,250) token 0 = subject
    keep["h0"] = h
    h = _drop(h, p_enc, train, masks, "embed")

    # --- A2: attention (SelfAttention_Family.py:194-213, 56-75)
    L = h.shape[1]
    q = F.linear(h, P[lay + "attention.query_projection.weight"], P[lay + "attention.query_projection.bias"])
    k = F.linear(h, P[lay + "attention.key_projection.weight"], P[lay + "attention.key_projection.bias"])
    v = F.linear(h, P[lay + "attention.value_projection.weight"], P[lay + "attention.value_projection.bias"])
    q = q.view(B, L, N_HEADS, D_HEAD).permute(0, 2, 1, 3)
    k = k.view(B, L, N_HEADS, D_HEAD).permute(0, 2, 1, 3)
    v = v.view(B, L, N_HEADS, D_HEAD).permute(0, 2, 1, 3)
    scores = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(D_HEAD))
    A = torch.softmax(scores, dim=-1)
    keep["attn_prob"] = A
    A = _drop(A, p_enc, train, masks, "attn")
    ao = (A @ v).permute(0, 2, 1, 3).reshape(B, L, N_HEADS * D_HEAD)           # (B,64,248)
    keep["attn_ctx"] = ao
    a1 = F.linear(ao, P[lay + "attention.out_projection.weight"], P[lay + "attention.out_projection.bias"])

    # --- A3: encoder layer (Transformer_EncDec.py:45-51) + final norm (:77-78)
    r1 = h + _drop(a1, p_enc, train, masks, "attn_out")
    n1 = F.layer_norm(r1, (D_MODEL,), P[lay + "norm1.weight"], P[lay + "norm1.bias"], LN_EPS)
    keep["n1"] = n1
    f1 = F.linear(n1, P[lay + "conv1.weight"][:, :, 0], P[lay + "conv1.bias"])   # Conv1d k=1 == Linear
    g1 = _drop(F.gelu(f1), p_enc, train, masks, "ffn_act")
    f2 = F.linear(g1, P[lay + "conv2.weight"][:, :, 0], P[lay + "conv2.bias"])
    r2 = n1 + _drop(f2, p_enc, train, masks, "ffn_out")
    n2 = F.layer_norm(r2, (D_MODEL,), P[lay + "norm2.weight"], P[lay + "norm2.bias"], LN_EPS)
    n3 = F.layer_norm(n2, (D_MODEL,), P["encoder.encoder.norm.weight"], P["encoder.encoder.norm.bias"], LN_EPS)
    keep["enc_out"] = n3

    # --- A4: keep tokens 0..62 = subject token + channels 0..61 (ATMS_retrieval.py:91)
    t = n3[:, :63, :]

    # --- A5: tsconv (ATMS_retrieval.py:102-109) + projection (:113-114) + flatten (:145)
    ts = "enc_eeg.0.tsconv."
    y = F.conv2d(t.unsqueeze(1), P[ts + "0.weight"], P[ts + "0.bias"])         # (B,40,63,226)
    y = F.avg_pool2d(y, (1, 51), (1, 5))                                        # (B,40,63,36)
    keep["conv1_pool"] = y
    if train:
        y, m1, v1 = _bn_train(y, P[ts + "2.weight"], P[ts + "2.bias"], (0, 2, 3))
        keep["bn1_mean"], keep["bn1_var"] = m1, v1
    else:
        y = _bn_eval(y, P[ts + "2.weight"], P[ts + "2.bias"], P[ts + "2.running_mean"], P[ts + "2.running_var"])
    y = F.elu(y)
    y = F.conv2d(y, P[ts + "4.weight"], P[ts + "4.bias"])                       # (B,40,1,36)
    keep["conv2"] = y
    if train:
        y, m2, v2 = _bn_train(y, P[ts + "5.weight"], P[ts + "5.bias"], (0, 2, 3))
        keep["bn2_mean"], keep["bn2_var"] = m2, v2
    else:
        y = _bn_eval(y, P[ts + "5.weight"], P[ts + "5.bias"], P[ts + "5.running_mean"], P[ts + "5.running_var"])
    y = F.elu(y)
    y = _drop(y, p_conv, train, masks, "conv")
    y = F.conv2d(y, P["enc_eeg.0.projection.0.weight"], P["enc_eeg.0.projection.0.bias"])   # 1x1
    feat = y.permute(0, 2, 3, 1).reshape(B, -1)                                 # 'b e h w -> b (h w) e' -> (B,1440)
    keep["feat"] = feat

    # --- A6: Proj_eeg (ATMS_retrieval.py:157-167)
    u = F.linear(feat, P["proj_eeg.0.weight"], P["proj_eeg.0.bias"])
    w = F.linear(F.gelu(u), P["proj_eeg.1.fn.1.weight"], P["proj_eeg.1.fn.1.bias"])
    s = u + _drop(w, p_proj, train, masks, "proj")
    out = F.layer_norm(s, (s.shape[-1],), P["proj_eeg.2.weight"], P["proj_eeg.2.bias"], LN_EPS)
    return out


def bn_running_update(running_mean, running_var, batch_mean, batch_var_biased, n, momentum=BN_MOMENTUM):
    """PyTorch BatchNorm running-stat update: unbiased variance, momentum 0.1."""
    unbiased = batch_var_biased * (n / (n - 1.0))
    return ((1 - momentum) * running_mean + momentum * batch_mean,
            (1 - momentum) * running_var + momentum * unbiased)


def fused_temporal_filter(w25):
    """(40,25) temporal taps -> (40,75) taps of the equivalent stride-5 conv that folds the
    AvgPool(1x51, stride 5) into the conv: weff[c,u] = 1/51 * sum_{t} w[c,t],
    over t in [max(0,u-50), min(24,u)].  Used to check the fused HIP kernel's tap table."""
    C, T = w25.shape
    weff = torch.zeros(C, T + 50, dtype=w25.dtype)
    for t in range(T):
        weff[:, t:t + 51] += w25[:, t:t + 1] / 51.0
    return weff


def state_spec(joint_train=False, num_subjects=2):
    """(key, shape, kind) for every entry of the reference ``ATMS().state_dict()``
    (pinned by tests/golden/atms_keys.json).  kind drives the synthetic weight recipe.
    joint_train / num_subjects: the ATMS of Retrieval/ATMS_retrieval_joint_train.py:172-192 (num_subjects = 10 there: one value-embedding
    Linear per subject when joint_train, and as many dead subject_wise_linear layers; pinned by tests/golden/joint_keys.json)."""
    e, l, ts = "encoder.enc_embedding.", "encoder.encoder.attn_layers.0.", "enc_eeg.0.tsconv."
    S = [("logit_scale", (), "logit_scale"),
         (e + "mask_token", (1, 250), "token")]
    if joint_train:
        for i in range(num_subjects):
            S += [(e + f"value_embedding.{i}.weight", (250, 250), "w"), (e + f"value_embedding.{i}.bias", (250,), "b")]
    else:
        S += [(e + "value_embedding.weight", (250, 250), "w"), (e + "value_embedding.bias", (250,), "b")]
    S += [(e + "position_embedding.pe", (1, 5000, 250), "pe"),
         (e + "temporal_embedding.embed.weight", (250, 4), "w"),
         (e + "subject_embedding.shared_embedding", (1, 250), "token"),
         (e + "subject_embedding.mask_embedding", (1, 250), "token"),
         (e + "subject_embedding.subject_embedding.weight", (10, 250), "token")]
    for nm, shp in (("query", (248, 250)), ("key", (248, 250)), ("value", (248, 250)), ("out", (250, 248))):
        S += [(l + f"attention.{nm}_projection.weight", shp, "w"), (l + f"attention.{nm}_projection.bias", (shp[0],), "b")]
    S += [(l + "conv1.weight", (256, 250, 1), "w"), (l + "conv1.bias", (256,), "b"),
          (l + "conv2.weight", (250, 256, 1), "w"), (l + "conv2.bias", (250,), "b"),
          (l + "norm1.weight", (250,), "g"), (l + "norm1.bias", (250,), "b"),
          (l + "norm2.weight", (250,), "g"), (l + "norm2.bias", (250,), "b"),
          ("encoder.encoder.norm.weight", (250,), "g"), ("encoder.encoder.norm.bias", (250,), "b")]
    for i in range(num_subjects):
        S += [(f"subject_wise_linear.{i}.weight", (250, 250), "w"), (f"subject_wise_linear.{i}.bias", (250,), "b")]
    S += [(ts + "0.weight", (40, 1, 1, 25), "w"), (ts + "0.bias", (40,), "b"),
          (ts + "2.weight", (40,), "g"), (ts + "2.bias", (40,), "b"),
          (ts + "2.running_mean", (40,), "rm"), (ts + "2.running_var", (40,), "rv"),
          (ts + "2.num_batches_tracked", (), "nbt"),
          (ts + "4.weight", (40, 40, 63, 1), "w"), (ts + "4.bias", (40,), "b"),
          (ts + "5.weight", (40,), "g"), (ts + "5.bias", (40,), "b"),
          (ts + "5.running_mean", (40,), "rm"), (ts + "5.running_var", (40,), "rv"),
          (ts + "5.num_batches_tracked", (), "nbt"),
          ("enc_eeg.0.projection.0.weight", (40, 40, 1, 1), "w"), ("enc_eeg.0.projection.0.bias", (40,), "b"),
          ("proj_eeg.0.weight", (1024, 1440), "w"), ("proj_eeg.0.bias", (1024,), "b"),
          ("proj_eeg.1.fn.1.weight", (1024, 1024), "w"), ("proj_eeg.1.fn.1.bias", (1024,), "b"),
          ("proj_eeg.2.weight", (1024,), "g"), ("proj_eeg.2.bias", (1024,), "b")]
    return S
