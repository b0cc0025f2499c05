"""MI355X parity of the HBM-resident input pipeline against tests/golden/dataset.npz, recorded from the reference's
Retrieval/eegdatasets_leaveone.py:EEGDataset on the same synthetic THINGS-EEG tree (tests/golden/make_golden_dataset.py), and against the oracle."""
import os

import numpy as np
import pytest
import torch

from eeg_image_decode_amd import synthetic as syn
from test_oracle_golden import DATASET_CONFIGS, JOINT_DATASET_CONFIGS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def things_tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("things_eeg"))
    return root, syn.write_things_eeg_tree(root, 20260927)


@pytest.mark.parametrize("name", list(DATASET_CONFIGS) + list(JOINT_DATASET_CONFIGS))
def test_device_dataset_matches_reference_fixture(name, things_tree, golden):
    joint = name in JOINT_DATASET_CONFIGS
    g = golden("dataset_joint.npz" if joint else "dataset.npz")
    root, cfg = things_tree
    if joint:                                                   # Retrieval/eegdatasets_joint_subjects.py: adap_subject instead of exclude_subject
        from eeg_image_decode_amd.datasets_joint import EEGDataset
        kw = dict(JOINT_DATASET_CONFIGS[name])
        kw["adap_subject"] = kw.pop("exclude_subject")
    else:
        from eeg_image_decode_amd.datasets import EEGDataset
        kw = dict(DATASET_CONFIGS[name])
    train = kw["train"]
    ds = EEGDataset(cfg["data_path"], config=cfg, features_dir=root, **kw)
    assert ds.data.is_cuda and len(ds) == int(g[f"{name}:len"]) and list(ds.data.shape) == g[f"{name}:data_shape"].tolist()
    assert np.array_equal(ds.labels.cpu().numpy(), g[f"{name}:labels"])
    assert len(ds.text) == int(g[f"{name}:n_text"]) and len(ds.img) == int(g[f"{name}:n_img"])
    idx = g[f"{name}:idx"]
    saved = torch.load(os.path.join(root, f"ViT-H-14_features_{'train' if train else 'test'}.pt"))
    for j, i in enumerate(idx.tolist()):
        x, label, text, tf, img, imf = ds[i]
        if train:
            assert np.array_equal(x.cpu().numpy(), g[f"{name}:x"][j])                        # cast + window: bit exact
        else:
            np.testing.assert_allclose(x.cpu().numpy(), g[f"{name}:x"][j], atol=1e-6)        # + mean over repetitions
        assert int(label) == int(g[f"{name}:label"][j]) and text == str(g[f"{name}:text"][j])
        assert os.path.relpath(img, root) == str(g[f"{name}:img"][j])
        assert torch.equal(tf.cpu(), saved["text_features"][int(g[f"{name}:text_row"][j])])
        assert torch.equal(imf.cpu(), saved["img_features"][int(g[f"{name}:img_row"][j])])
    assert abs(float(ds.data.double().sum()) - float(g[f"{name}:data_sum"])) < 1e-3
    if not train:
        np.testing.assert_allclose(ds.data.cpu().numpy(), g[f"{name}:data"], atol=1e-6)
    # the loader == DataLoader(dataset, ...) semantics: every sample once per epoch, batches are rows of the resident tensors
    ld = ds.loader(batch_size=256, shuffle=True, drop_last=False, generator=torch.Generator().manual_seed(3))
    order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(3))
    count = 0
    for bi, (x, y, text, tf, img, imf) in enumerate(ld):
        sel = order[bi * 256:bi * 256 + 256].cuda()
        assert torch.equal(x, ds.data[sel]) and torch.equal(y, ds.labels[sel])
        ti, ii = ds._rows(sel.cpu().numpy())
        assert torch.equal(tf, ds.text_features[torch.from_numpy(ti).cuda()]) and torch.equal(imf, ds.img_features[torch.from_numpy(ii).cuda()])
        assert text[0] == ds.text[int(ti[0])] and img[-1] == ds.img[int(ii[-1])]
        count += len(y)
        if bi >= 20:
            break
    assert count == min(len(ds), 21 * 256)


def test_loader_feeds_train_and_evaluate_loops(tmp_path):
    """end to end in the reference's shapes: a (small) tree with 63 channels x 250 samples -> EEGDataset -> loader -> train_model / evaluate_model"""
    from eeg_image_decode_amd import optim, retrieval
    from eeg_image_decode_amd.atms import ATMS
    from eeg_image_decode_amd.datasets import EEGDataset
    root = str(tmp_path)
    cfg = syn.write_things_eeg_tree(root, 11, subjects=("sub-01",), channels=63, n_times=300, dt=0.004, train_classes=12, test_classes=200, test_reps=3)
    tr = EEGDataset(cfg["data_path"], subjects=["sub-01"], train=True, config=cfg, features_dir=root)
    te = EEGDataset(cfg["data_path"], subjects=["sub-01"], train=False, config=cfg, features_dir=root)
    assert tuple(tr.data.shape) == (12 * 10 * 4, 63, 250) and tuple(te.data.shape) == (200, 63, 250)
    torch.manual_seed(0)
    m = ATMS().cuda()
    opt = optim.AdamW(m.parameters(), lr=3e-4)
    loss, acc, feats = retrieval.train_model("sub-01", m, tr.loader(batch_size=96, shuffle=True, drop_last=True), opt, "cuda", tr.text_features,
                                             tr.img_features, None)
    assert np.isfinite(loss) and 0.0 <= acc <= 1.0 and tuple(feats.shape) == (480, 1024)
    l2, a2, top5 = retrieval.evaluate_model("sub-01", m, te.loader(batch_size=1, shuffle=False), "cuda", te.text_features, te.img_features, 200, None)
    assert np.isfinite(l2) and 0.0 <= a2 <= top5 <= 1.0
