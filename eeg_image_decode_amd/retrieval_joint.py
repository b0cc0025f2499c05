"""Joint-subject ATM-S (Retrieval/ATMS_retrieval_joint_train.py): the same encoder with ONE VALUE EMBEDDING PER SUBJECT -- sample i goes through
value_embedding[str(subject_ids[i])] (models/subject_layers/Embed.py:127-131,142-144) -- trained on the pooled data of all subjects.

Same constructor as the reference's class (ATMS_retrieval_joint_train.py:172-181): ATMS(sequence_length=250, num_subjects=10, joint_train=False).
As there, num_subjects only sizes the (unused) subject_wise_linear list; the subject-token table and the per-subject embeddings always cover
subjects 0..9, and an id outside that range is an error in joint mode (a KeyError in the reference, EegclipError here).

The loops are the retrieval ones (the reference duplicates them in this script; its train / eval loops feed one subject id per call,
ATMS_retrieval_joint_train.py:219-222, which is the single-GEMM case of the engine; batches that mix subjects run one GEMM per subject over the
subject-ordered batch, see atms._Engine._joint_layout).
"""
from . import atms as _a
from .atms import Config, Enc_eeg, PatchEmbedding, Proj_eeg, iTransformer      # noqa: F401
from .retrieval import evaluate_model, extract_id_from_string, get_eegfeatures, main_train_loop, train_model      # noqa: F401


class ATMS(_a.ATMS):
    def __init__(self, sequence_length=250, num_subjects=10, joint_train=False):
        super().__init__(63, sequence_length, num_subjects, joint_train=joint_train, table_subjects=10)
