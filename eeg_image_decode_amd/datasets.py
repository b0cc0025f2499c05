"""The reference's input pipeline on the device: Retrieval/eegdatasets_leaveone.py:EEGDataset with
the same constructor, attributes, item tuple and index arithmetic, but the whole split lives in HBM (one subject = 66,160 x 63 x 250 float32 =
4.2 GB of 288 GB) and shuffled batches are assembled there.

  * on-disk format unchanged: <data_path>/<sub>/preprocessed_eeg_{training,test}.npy, a pickled dict {'preprocessed_eeg_data' (images, reps,
    channels, T) float64, 'ch_names', 'times'} (EEG-preprocessing/preprocessing_utils.py:240-300); cached CLIP features
    ViT-H-14_features_{train,test}.pt {'text_features', 'img_features'}; data_config.json for the image directories (texts and image paths
    come from the sorted directory listings, eegdatasets_leaveone.py:90-147);
  * staging: float64 chunks go host -> HBM and one HIP pass (eegclip_stage_eeg) applies the float32 cast, the time-window mask and, for the
    test split, the mean over the 80 repetitions (:157,:220,:293-306) -- the reference does this with torch on the host, holding the float64
    and float32 copies of every subject in RAM;
  * EEGDataset[i] returns the reference's (x, label, text, text_features, img, img_features) (:326-375) with device tensors;
  * EEGDataset.loader(batch_size, shuffle, drop_last) replaces DataLoader(dataset, ...): one gather launch per tensor per batch
    (eegclip_gather_rows) straight from the resident tensors -- no per-sample __getitem__, no collate, no H2D copy inside the step.  It yields
    what DataLoader's default collate yields (tensors + lists of str), so train_model / evaluate_model consume it unchanged.

The CLIP text / image encoders (open_clip ViT-H-14) are not part of this path: the cached feature files must exist (the reference writes them
on first use, :62-74); the `classes` / `pictures` subsets re-encode on every construction (:75-77) and are therefore not offered.
"""
import json
import os
import pickle

import numpy as np
import torch

from ._lib import EegclipError, check, lib, raw_stream, require_cuda

model_type = "ViT-H-14"                                   # eegdatasets_leaveone.py:17
_CHUNK_BYTES = 1 << 29                                    # float64 staging chunk (host -> HBM -> eegclip_stage_eeg)


def load_config(path="data_config.json"):
    """eegdatasets_leaveone.py:24-34 (the reference reads it at import time from the working directory)"""
    with open(path, "r") as f:
        return json.load(f)


def _stream():
    return raw_stream()


def _listing(directory):
    """texts and image paths from the sorted class directories (eegdatasets_leaveone.py:90-106,137-143)"""
    dirnames = sorted(d for d in os.listdir(directory) if os.path.isdir(os.path.join(directory, d)))
    texts = [f"This picture is {d[d.index('_') + 1:]}" for d in dirnames if "_" in d]
    images = []
    for d in dirnames:
        fp = os.path.join(directory, d)
        images.extend(os.path.join(fp, i) for i in sorted(i for i in os.listdir(fp) if i.lower().endswith((".png", ".jpg", ".jpeg"))))
    return texts, images


class EEGDataset:
    """One split (training or test) of one or several THINGS-EEG subjects, resident in HBM; constructor, attributes and item tuple of
    Retrieval/eegdatasets_leaveone.py:36-375 (keyword-only extras: config, features_dir, device)."""

    def __init__(self, data_path, exclude_subject=None, subjects=None, train=True, time_window=[0, 1.0], classes=None, pictures=None, val_size=None,
                 *, config=None, features_dir=".", device="cuda"):
        if classes is not None or pictures is not None:
            raise EegclipError("classes / pictures subsets run the CLIP encoders on every construction (eegdatasets_leaveone.py:75-77); "
                               "only the cached-feature configuration (classes=None, pictures=None) is on this path")
        self.data_path = data_path
        self.train = train
        self.subject_list = os.listdir(data_path)
        self.subjects = self.subject_list if subjects is None else subjects
        self.n_sub = len(self.subjects)
        self.time_window = time_window
        self.n_cls = 1654 if train else 200
        self.classes, self.pictures = classes, pictures
        self.exclude_subject = exclude_subject
        self.val_size = val_size
        self.device = torch.device(device)
        assert any(sub in self.subject_list for sub in self.subjects)
        cfg = config if config is not None else load_config()
        self.text, self.img = _listing(cfg["img_directory_training"] if train else cfg["img_directory_test"])
        self.data, self.labels = self._stage()
        fn = os.path.join(features_dir, f"{model_type}_features_train.pt" if train else f"{model_type}_features_test.pt")
        if not os.path.exists(fn):
            raise EegclipError(f"{fn} not found: the cached CLIP features are an input of this path (the reference creates them with open_clip "
                               "on first use, eegdatasets_leaveone.py:62-74)")
        saved = torch.load(fn)
        self.text_features = saved["text_features"].to(self.device).float().contiguous()
        self.img_features = saved["img_features"].to(self.device).float().contiguous()

    # ---- staging ------------------------------------------------------------------------------------------------------------------------------
    def _files(self):
        for sub in self.subjects:
            if self.train:
                if sub == self.exclude_subject:
                    continue
                yield os.path.join(self.data_path, sub, "preprocessed_eeg_training.npy")
            elif sub == self.exclude_subject or self.exclude_subject is None:
                yield os.path.join(self.data_path, sub, "preprocessed_eeg_test.npy")

    def _stage(self):
        dev = self.device
        require_cuda(torch.empty(0, device=dev), "EEGDataset(device=...) -- the split is staged in HBM, there is no host fallback")
        per_class, n_cls = (10, 1654) if self.train else (1, 200)
        blocks, labels = [], []
        L = lib()
        for path in self._files():
            with open(path, "rb") as f:
                d = pickle.load(f)                                    # == np.load(path, allow_pickle=True) on a pickle under an .npy name
            eeg = np.ascontiguousarray(d["preprocessed_eeg_data"][:n_cls * per_class], dtype=np.float64)
            self.times = torch.from_numpy(np.asarray(d["times"]))[50:]
            self.ch_names = d["ch_names"]
            n_items, reps, C, T = eeg.shape
            if len(self.times) != T:
                raise EegclipError(f"{path}: {T} stored samples but times[50:] has {len(self.times)} entries")
            start, end = self.time_window
            tidx = torch.nonzero((self.times >= start) & (self.times <= end)).flatten().to(torch.int32)
            Tw = int(tidx.numel())
            if Tw == 0:
                raise EegclipError(f"time_window {self.time_window} selects no sample")
            tidx_d = tidx.to(dev) if Tw != T else None             # whole window kept (the real data): the flat cast path
            rows_out = n_items if not self.train else n_items * reps
            out = torch.empty(rows_out, C, Tw, dtype=torch.float32, device=dev)
            step = max(1, _CHUNK_BYTES // (reps * C * T * 8))
            for i0 in range(0, n_items, step):
                n = min(step, n_items - i0)
                chunk = torch.from_numpy(eeg[i0:i0 + n]).to(dev)                                  # float64, resident only for this chunk
                dst = out[i0:i0 + n] if not self.train else out[i0 * reps:(i0 + n) * reps]
                check(L.eegclip_stage_eeg(chunk.data_ptr(), dst.data_ptr(), n, reps, C, T, tidx_d.data_ptr() if tidx_d is not None else None, Tw,
                                          0 if self.train else 1, _stream()),
                      "stage_eeg")
                del chunk
            blocks.append(out)
            lab = torch.arange(n_items // per_class, dtype=torch.long).repeat_interleave(per_class)          # :186,:217 class of every image
            labels.append(lab.repeat_interleave(4) if self.train else lab)                                   # :263 (the reference's fixed 4)
        if not blocks:
            raise EegclipError("no subject file selected (subjects / exclude_subject)")
        data = blocks[0] if len(blocks) == 1 else torch.cat(blocks, 0)
        print(f"Data tensor shape: {tuple(data.shape)}, label tensor shape: {(sum(len(l) for l in labels),)}, text length: {len(self.text)}, "
              f"image length: {len(self.img)}")
        return data, torch.cat(labels).to(dev)

    # ---- the reference's item access ------------------------------------------------------------------------------------------------------------
    def _rows(self, index):
        """text row / image row of sample(s) `index` (eegdatasets_leaveone.py:332-348); works on ints and integer arrays"""
        if self.train:
            r = index % (self.n_cls * 10 * 4)
            return r // 40, r // 4
        r = index % (self.n_cls * 1 * 80)
        return r, r

    def __getitem__(self, index):
        ti, ii = self._rows(int(index))
        return self.data[index], self.labels[index], self.text[ti], self.text_features[ti], self.img[ii], self.img_features[ii]

    def __len__(self):
        return self.data.shape[0]

    def loader(self, batch_size=1, shuffle=False, drop_last=False, generator=None):
        return DeviceLoader(self, batch_size, shuffle, drop_last, generator)


class DeviceLoader:
    """DataLoader(dataset, batch_size, shuffle, drop_last) over the HBM-resident split: same batch tuple as the default collate."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, generator=None):
        self.dataset, self.batch_size, self.shuffle, self.drop_last, self.generator = dataset, int(batch_size), shuffle, drop_last, generator
        row = dataset.data.shape[1] * dataset.data.shape[2]
        if row % 2:
            raise EegclipError("a sample must hold an even number of floats (8-byte gather pieces); 63 x 250 does")

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    @staticmethod
    def _gather(src, idx_dev, n):
        row_floats = src[0].numel() * (2 if src.dtype == torch.long else 1)           # an int64 label travels as one 8-byte piece
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        check(lib().eegclip_gather_rows(out.data_ptr(), row_floats, src.data_ptr(), row_floats, idx_dev.data_ptr(), n, row_floats, 0, _stream()),
              "gather_rows")
        return out

    def __iter__(self):
        ds = self.dataset
        n = len(ds)
        order = torch.randperm(n, generator=self.generator).numpy() if self.shuffle else np.arange(n)       # host order, like DataLoader's sampler
        dev = ds.data.device
        for i0 in range(0, n, self.batch_size):
            idx = order[i0:i0 + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            ti, ii = ds._rows(idx)
            both = torch.from_numpy(np.stack([idx, ti, ii]).astype(np.int32)).to(dev, non_blocking=True)    # one small H2D per batch
            b = len(idx)
            yield (self._gather(ds.data, both[0], b), self._gather(ds.labels, both[0], b), [ds.text[j] for j in ti],
                   self._gather(ds.text_features, both[1], b), [ds.img[j] for j in ii], self._gather(ds.img_features, both[2], b))
