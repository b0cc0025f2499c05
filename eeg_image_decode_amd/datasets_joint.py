"""The joint-subject variant of the input pipeline (Retrieval/eegdatasets_joint_subjects.py): the same on-disk format, staging and loader as
datasets.EEGDataset, with that module's constructor -- `adap_subject` instead of `exclude_subject`: every listed subject contributes training
data (:153-154), the test split is the adaptation subject's (:195)."""
from . import datasets as _d
from .datasets import DeviceLoader, load_config, model_type      # noqa: F401


class EEGDataset(_d.EEGDataset):
    def __init__(self, data_path, adap_subject=None, subjects=None, train=True, time_window=[0, 1.0], classes=None, pictures=None, *, config=None,
                 features_dir=".", device="cuda"):
        self.adap_subject = adap_subject
        super().__init__(data_path, exclude_subject=adap_subject, subjects=subjects, train=train, time_window=time_window, classes=classes,
                         pictures=pictures, config=config, features_dir=features_dir, device=device)

    def _files(self):
        import os
        for sub in self.subjects:
            if self.train:
                yield os.path.join(self.data_path, sub, "preprocessed_eeg_training.npy")
            elif sub == self.adap_subject or self.adap_subject is None:
                yield os.path.join(self.data_path, sub, "preprocessed_eeg_test.npy")
