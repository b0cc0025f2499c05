"""CPU restatement of the reference's input pipeline -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Follows Retrieval/eegdatasets_leaveone.py:EEGDataset (the class behind SURVEY.md section 8a row D and section 8f row 4):
  * on-disk format (:151-156,:199-203; written by EEG-preprocessing/preprocessing_utils.py:240-300): a pickled dict under an .npy name,
    {'preprocessed_eeg_data' (images, repetitions, channels, T) float64, 'ch_names', 'times'}; `times[50:]` labels the T stored samples;
  * load_data (:79-291): texts "This picture is <dir name after the first '_'>" from the sorted class directories, image paths from the sorted
    listings, per-subject trial blocks -- training: every repetition is a sample, order ((class, image), repetition), the held-out subject is
    skipped; test: the mean over repetitions (taken AFTER the float32 cast), only the held-out subject (or every subject when none is held out);
  * extract_eeg (:293-306): boolean mask (times >= start) & (times <= end) on the last axis;
  * __getitem__ (:326-375): text row = (index mod per-subject length) // 40, image row = ... // 4 (training); index mod 16000 (test: n_cls*80,
    which never wraps for the 200 averaged items).
Pinned by tests/golden/dataset.npz, produced by the reference class itself on eeg_image_decode_amd.synthetic.write_things_eeg_tree.
"""
import os
import pickle

import numpy as np


def read_subject_file(path):
    """np.load(path, allow_pickle=True) of the reference (:155,:201) on a pickle-under-.npy file = pickle.load"""
    with open(path, "rb") as f:
        return pickle.load(f)


def list_texts_and_images(directory, classes=None, pictures=None):
    """eegdatasets_leaveone.py:90-147"""
    dirnames = sorted(d for d in os.listdir(directory) if os.path.isdir(os.path.join(directory, d)))
    tdirs = [dirnames[i] for i in classes] if classes is not None else dirnames
    texts = [f"This picture is {d[d.index('_') + 1:]}" for d in tdirs if "_" in d]
    images = []

    def listing(folder):
        fp = os.path.join(directory, folder)
        return [os.path.join(fp, i) for i in sorted(i for i in os.listdir(fp) if i.lower().endswith((".png", ".jpg", ".jpeg")))]

    if classes is not None and pictures is not None:
        for c, p in zip(classes, pictures):
            if c < len(dirnames):
                li = listing(dirnames[c])
                if p < len(li):
                    images.append(li[p])
    elif classes is not None:
        for c in classes:
            if c < len(dirnames):
                images.extend(listing(dirnames[c]))
    else:
        for d in dirnames:
            images.extend(listing(d))
    return texts, images


def load_split(data_path, img_dir, subjects, exclude_subject=None, train=True, time_window=(0, 1.0), classes=None, pictures=None, joint=False):
    """-> data (n, C, Tw) float32, labels (n,) int64, texts, images, times (after [50:]), ch_names
    joint=True: Retrieval/eegdatasets_joint_subjects.py, whose `adap_subject` (passed here as exclude_subject) never drops a TRAINING subject
    (:153-154 commented out) and selects the test subject exactly like exclude_subject does (:195)."""
    texts, images = list_texts_and_images(img_dir, classes, pictures)
    blocks, labels = [], []
    times = ch_names = None
    for sub in subjects:
        if train:
            if sub == exclude_subject and not joint:
                continue
            d = read_subject_file(os.path.join(data_path, sub, "preprocessed_eeg_training.npy"))
            eeg = d["preprocessed_eeg_data"].astype(np.float32)
            times, ch_names = d["times"][50:], d["ch_names"]
            if classes is not None and pictures is not None:
                for c, p in zip(classes, pictures):
                    if c + p < len(eeg):
                        blocks.append(eeg[c + p:c + p + 1])
                        labels.append(np.full(1, c, np.int64))
            else:
                for c in (classes if classes is not None else range(1654)):
                    blocks.append(eeg[c * 10:c * 10 + 10])
                    labels.append(np.full(10, c, np.int64))
        else:
            if not (sub == exclude_subject or exclude_subject is None):
                continue
            d = read_subject_file(os.path.join(data_path, sub, "preprocessed_eeg_test.npy"))
            eeg = d["preprocessed_eeg_data"].astype(np.float32)
            times, ch_names = d["times"][50:], d["ch_names"]
            for c in range(200):
                if classes is not None and c not in classes:
                    continue
                blocks.append(eeg[c].mean(0, dtype=np.float32)[None])          # mean over repetitions in float32, like torch.mean
                labels.append(np.full(1, c, np.int64))
    cat = np.concatenate(blocks, 0)
    data = cat.reshape(-1, *cat.shape[2:]) if train else cat
    lab = np.concatenate(labels)
    if train:
        lab = np.repeat(lab, 4)
        if classes is not None:                       # remap to 0..len(classes)-1 in order of first appearance (:270-278)
            order = {v: i for i, v in enumerate(dict.fromkeys(lab.tolist()))}
            lab = np.array([order[v] for v in lab.tolist()], np.int64)
    mask = (times >= time_window[0]) & (times <= time_window[1])
    return data[..., mask], lab, texts, images, times, ch_names


def item_rows(index, train, n_cls, classes=None, pictures=None):
    """(text row, image row) of sample `index` (:332-366)"""
    k = len(classes) if classes is not None else n_cls
    per = 1 if pictures is not None else 10
    n_train, n_test = k * per * 4, k * 80
    if train:
        r = index % n_train
        return r // (per * 4), r // 4
    r = index % n_test
    return r, r
