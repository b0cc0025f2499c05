"""Deterministic synthetic inputs and weights (numpy Philox; independent of torch's RNG).

There is no THINGS-EEG data and no CLIP checkpoint on the build or GPU box, so every test,
fixture and benchmark uses the stand-ins below:

  * EEG              x ~ N(0,1) f32 (B,63,250)   (real data is MVNN-whitened, ~unit variance)
  * image/text feats unit-norm N(0,1) f32 (.,1024)  (Retrieval/eegdatasets_leaveone.py:304,318 L2-normalises)
  * weights          a per-key Philox stream scaled like a trained network (NOT torch's default init,
                     so fixtures do not depend on the torch build).

Host-side helpers only: nothing here is on the compute path.
"""
import math
import zlib

import numpy as np


def _rng(seed, tag):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(tag.encode())]))


def eeg_batch(seed, batch, channels=63, time=250):
    return _rng(seed, "eeg").standard_normal((batch, channels, time), dtype=np.float32)


def unit_features(seed, n, dim=1024, tag="img"):
    f = _rng(seed, "feat:" + tag).standard_normal((n, dim), dtype=np.float32)
    return (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)


def sinusoid_table(n_pos, d_model):
    """Same closed form as the reference buffer `position_embedding.pe` (Embed.py:12-20), in f32."""
    pos = np.arange(n_pos, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d_model))).astype(np.float32)
    pe = np.zeros((n_pos, d_model), dtype=np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe


def make_tensor(seed, key, shape, kind):
    """One synthetic state_dict entry.  kind: w|b|g|token|pe|rm|rv|nbt|logit_scale."""
    r = _rng(seed, key)
    shape = tuple(shape)
    if kind == "w":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        return (r.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(fan_in))).astype(np.float32)
    if kind == "b":
        return (0.05 * r.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if kind == "g":
        return (1.0 + 0.1 * r.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if kind == "token":
        return r.standard_normal(shape, dtype=np.float32)
    if kind == "pe":
        return sinusoid_table(shape[-2], shape[-1]).reshape(shape)
    if kind == "rm":
        return (0.1 * r.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if kind == "rv":
        return (1.0 + 0.2 * r.random(shape, dtype=np.float32)).astype(np.float32)
    if kind == "nbt":
        return np.zeros(shape, dtype=np.int64)
    if kind == "logit_scale":
        return np.asarray(math.log(1 / 0.07), dtype=np.float32).reshape(shape)
    raise ValueError(kind)


def make_state(seed, spec):
    """spec: iterable of (key, shape, kind) -> {key: ndarray}."""
    return {k: make_tensor(seed, k, s, kind) for k, s, kind in spec}


def learnable_pairs(seed, n_classes, per_class, noise=1.0, dim=1024, channels=63, time=250, mix_rank=64):
    """Class-structured EEG/target pairs for accuracy tests: class prototypes
    P (n_classes, dim) unit-norm; EEG for class c = reshape(M . P_c) + noise * N(0,1) with a fixed random
    low-rank mixing M, so a contrastive encoder can learn the mapping in a few hundred steps.
    Returns (eeg (n,63,250) f32, labels (n,) i64, prototypes (n_classes, dim) f32)."""
    protos = unit_features(seed, n_classes, dim, tag="proto")
    r = _rng(seed, "mix")
    m1 = r.standard_normal((dim, mix_rank), dtype=np.float32) / np.float32(math.sqrt(dim))
    m2 = r.standard_normal((mix_rank, channels * time), dtype=np.float32) / np.float32(math.sqrt(mix_rank))
    labels = np.repeat(np.arange(n_classes, dtype=np.int64), per_class)
    base = (protos @ m1) @ m2 * np.float32(math.sqrt(dim))          # ~unit variance per element
    eeg = base[labels] + noise * _rng(seed, "pairnoise").standard_normal((labels.size, channels * time), dtype=np.float32)
    return eeg.reshape(-1, channels, time).astype(np.float32), labels, protos
