"""eeg_image_decode_amd -- MI355X-native (gfx950) hot path of dongyangli-del/EEG_Image_decode.

Public surface mirrors the reference's Python API for the contrastive-training / diffusion-prior
path.  All arithmetic runs in hand-written HIP kernels reached through the
C-ABI in include/eegclip.h; there is NO CPU fallback: importing the compute modules on a machine
without the built library (or calling them without a GPU) raises.
"""
__version__ = "0.1.0"
