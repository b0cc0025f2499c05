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


def write_things_eeg_tree(root, seed, subjects=("sub-01", "sub-02"), channels=4, n_times=110, train_classes=1654, imgs_per_class=10,
                          train_reps=4, test_classes=200, test_reps=5, feat_dim=1024, dt=0.01):
    """A synthetic stand-in for the THINGS-EEG2 tree the reference's EEGDataset reads (Retrieval/eegdatasets_leaveone.py:24-34,151-156,
    199-203, preprocessing_utils.py:240-300), in the reference's ON-DISK FORMAT, with small channel / time extents so it is cheap to write:

      <root>/data/<sub>/preprocessed_eeg_training.npy   pickled dict {'preprocessed_eeg_data' (classes*imgs, reps, C, T) float64,
                                                                      'ch_names' list[str], 'times' (n_times,) float64}
      <root>/data/<sub>/preprocessed_eeg_test.npy       same keys, (test_classes, test_reps, C, T)
      <root>/images/training_images/00001_<name>/<name>_01b.jpg ...   (empty files: only the directory listing is read when the cached
      <root>/images/test_images/00001_<name>/...                       features exist)
      <root>/ViT-H-14_features_{train,test}.pt          {'text_features' (classes, D), 'img_features' (classes*imgs, D)} float32
      <root>/data_config.json                           {'data_path', 'img_directory_training', 'img_directory_test'}

    `times` has n_times entries `dt` apart with times[50] = 0 s; the reference drops the first 50 (`times[50:]`), so the stored EEG has
    T = n_times - 50 samples (the real data: 250 Hz, n_times = 300, dt = 0.004 -> 250 samples in [0, 1] s).  Returns the dict written to
    data_config.json."""
    import json
    import os
    import pickle

    import torch
    T = n_times - 50
    times = np.round(-0.2 + dt * np.arange(n_times), 10) - (0.3 if dt == 0.01 else 50 * dt - 0.2)          # times[50:] starts at 0.0
    ch_names = [f"CH{i}" for i in range(channels)]
    data_dir = os.path.join(root, "data")
    for sub in subjects:
        os.makedirs(os.path.join(data_dir, sub), exist_ok=True)
        for split, shape in (("training", (train_classes * imgs_per_class, train_reps, channels, T)), ("test", (test_classes, test_reps, channels, T))):
            arr = _rng(seed, f"eegfile:{sub}:{split}").standard_normal(shape)                 # float64, like the MVNN-whitened data
            with open(os.path.join(data_dir, sub, f"preprocessed_eeg_{split}.npy"), "wb") as f:
                # a plain pickle (protocol 4) under an .npy name, as EEG-preprocessing/preprocessing_utils.py:254-257,295-299 writes it:
                # np.load(allow_pickle=True) falls back to pickle.load for a file without the npy magic and returns the dict itself
                pickle.dump({"preprocessed_eeg_data": arr, "ch_names": ch_names, "times": times}, f, protocol=4)
    cfg = {"data_path": data_dir, "img_directory_training": os.path.join(root, "images", "training_images"),
           "img_directory_test": os.path.join(root, "images", "test_images")}
    for key, n_cls, per in (("img_directory_training", train_classes, imgs_per_class), ("img_directory_test", test_classes, 1)):
        for c in range(n_cls):
            name = f"thing{c:04d}" + ("_big" if c % 7 == 0 else "")            # some names carry a second '_' (only the first one splits)
            d = os.path.join(cfg[key], f"{c + 1:05d}_{name}")
            os.makedirs(d, exist_ok=True)
            for j in range(per):
                open(os.path.join(d, f"{name}_{j + 1:02d}{'s' if key.endswith('test') else 'b'}.jpg"), "wb").close()
    for split, n_cls, per in (("train", train_classes, imgs_per_class), ("test", test_classes, 1)):
        torch.save({"text_features": torch.from_numpy(unit_features(seed, n_cls, feat_dim, tag=f"ds-text-{split}")),
                    "img_features": torch.from_numpy(unit_features(seed, n_cls * per, feat_dim, tag=f"ds-img-{split}"))},
                   os.path.join(root, f"ViT-H-14_features_{split}.pt"))
    with open(os.path.join(root, "data_config.json"), "w") as f:
        json.dump(cfg, f)
    return cfg
