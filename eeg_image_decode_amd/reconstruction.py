"""Reconstruction-objective training of the same ATM-S encoder (Generation/ATMS_reconstruction.py:191-249): the embeddings that
feed the diffusion prior are trained with  10 * (0.9 * MSE(z, z_img) + 0.1 * ClipLoss(z, z_img))  instead of the retrieval
script's 0.99 / 0.01 image / text InfoNCE mix.  Model, evaluation loop and epoch driver are the retrieval ones (the reference
duplicates them verbatim in both scripts); only the per-batch objective differs.  Same signatures as the reference.
"""
from .atms import ATMS, Config, Enc_eeg, PatchEmbedding, Proj_eeg, iTransformer      # noqa: F401  (same classes in both reference scripts)
from .retrieval import evaluate_model, extract_id_from_string, get_eegfeatures, main_train_loop      # noqa: F401
from . import retrieval as _r


def train_model(sub, eeg_model, dataloader, optimizer, device, text_features_all, img_features_all, config):
    """one epoch; returns (average_loss, accuracy, features) like ATMS_reconstruction.py:191-249 (alpha = 0.90 hard-coded there)"""
    return _r.train_model(sub, eeg_model, dataloader, optimizer, device, text_features_all, img_features_all, config,
                          objective="reconstruction", alpha=0.90)
