#!/usr/bin/env python3
"""Golden fixtures for the input pipeline: runs THE REFERENCE's
Retrieval/eegdatasets_leaveone.py:EEGDataset (imported in place; clip / open_clip / torchvision stubbed -- the cached-feature path never calls
them) on the synthetic THINGS-EEG tree of eeg_image_decode_amd.synthetic.write_things_eeg_tree and stores OUTPUTS only:

    tests/golden/dataset.npz, dataset_joint.npz (the joint-subject variant, eegdatasets_joint_subjects.py)     per configuration: len, the whole label tensor, and for a spread of indices the item tuple
                                 (EEG window, label, text, image path relative to the tree, rows of the feature tables)

    python tests/golden/make_golden_dataset.py
"""
import importlib.util
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
from eeg_image_decode_amd import synthetic as syn  # noqa: E402

SEED = 20260927
JOINT_CONFIGS = {
    # Retrieval/eegdatasets_joint_subjects.py: adap_subject never drops a training subject; it selects the test subject
    "joint_train_adapt_sub02": dict(subjects=["sub-01", "sub-02"], adap_subject="sub-02", train=True),
    "joint_test_adapt_sub02": dict(subjects=["sub-01", "sub-02"], adap_subject="sub-02", train=False),
    # (adap_subject=None with several subjects concatenates their test sets, but __getitem__ indexes the 200 texts with index % 16000 and
    #  raises IndexError from item 200 on -- in both dataset modules; not a usable configuration)
}
CONFIGS = {
    # name: EEGDataset kwargs (data_path is filled in)
    "train_two_subjects": dict(subjects=["sub-01", "sub-02"], train=True),
    "train_leave_sub02_out": dict(subjects=["sub-01", "sub-02"], exclude_subject="sub-02", train=True),
    "test_sub01": dict(subjects=["sub-01"], train=False),
    "test_leave_sub02_out": dict(subjects=["sub-01", "sub-02"], exclude_subject="sub-02", train=False),
    "train_window": dict(subjects=["sub-02"], train=True, time_window=[0.1, 0.35]),
}


def probe_indices(n):
    idx = sorted({0, 1, 3, 4, 39, 40, 41, 399, 400, n // 2 - 1, n // 2, n // 2 + 43, n - 41, n - 2, n - 1} & set(range(n)))
    return idx


def main():
    root = tempfile.mkdtemp(prefix="things_eeg_")
    try:
        syn.write_things_eeg_tree(root, SEED)
        for name in ("clip", "open_clip", "torchvision", "torchvision.transforms"):
            sys.modules[name] = types.ModuleType(name)
        sys.modules["open_clip"].create_model_and_transforms = lambda *a, **k: (None, None, None)
        sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
        os.chdir(root)                                   # the reference opens data_config.json and the feature caches relative to the cwd
        spec = importlib.util.spec_from_file_location("ref_ds", os.path.join(REF, "Retrieval", "eegdatasets_leaveone.py"))
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        spec = importlib.util.spec_from_file_location("ref_ds_joint", os.path.join(REF, "Retrieval", "eegdatasets_joint_subjects.py"))
        refj = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(refj)
        for which, (mod, configs) in {"dataset.npz": (ref, CONFIGS), "dataset_joint.npz": (refj, JOINT_CONFIGS)}.items():
            record(mod, configs, root, os.path.join(HERE, which))
    finally:
        os.chdir(HERE)
        shutil.rmtree(root, ignore_errors=True)


def record(ref, configs, root, path):
    if True:
        out = {}
        for name, kw in configs.items():
            ds = ref.EEGDataset(ref.data_path, **kw)
            n = len(ds)
            out[f"{name}:len"] = np.int64(n)
            out[f"{name}:data_shape"] = np.array(ds.data.shape)
            out[f"{name}:labels"] = ds.labels.numpy().astype(np.int32)
            out[f"{name}:n_text"], out[f"{name}:n_img"] = np.int64(len(ds.text)), np.int64(len(ds.img))
            idx = probe_indices(n)
            out[f"{name}:idx"] = np.array(idx)
            items = [ds[i] for i in idx]
            out[f"{name}:x"] = np.stack([it[0].numpy() for it in items])
            out[f"{name}:label"] = np.array([int(it[1]) for it in items])
            out[f"{name}:text"] = np.array([it[2] for it in items])
            out[f"{name}:img"] = np.array([os.path.relpath(it[4], root) for it in items])
            tf, imf = ds.text_features.numpy(), ds.img_features.numpy()
            out[f"{name}:text_row"] = np.array([int(np.flatnonzero((tf == it[3].numpy()).all(1))[0]) for it in items])
            out[f"{name}:img_row"] = np.array([int(np.flatnonzero((imf == it[5].numpy()).all(1))[0]) for it in items])
            if not kw["train"]:
                out[f"{name}:data"] = ds.data.numpy()                  # the averaged test set is small: keep all of it
            out[f"{name}:data_sum"] = np.float64(ds.data.double().sum().item())
        np.savez_compressed(path, **out)
        print("wrote", os.path.basename(path), {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith(("len", "data_shape"))})


if __name__ == "__main__":
    main()
