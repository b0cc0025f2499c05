"""CPU-side checks of the host layer: reference-compatible state_dict, C-ABI symbol coverage, fail-loudly behaviour."""
import ctypes
import json
import os
import re
import subprocess

import pytest
import torch

from conftest import GOLDEN, ROOT


def test_atms_state_dict_matches_reference_keys_and_shapes():
    from eeg_image_decode_amd.atms import ATMS
    m = ATMS()
    with open(os.path.join(GOLDEN, "atms_keys.json")) as f:
        ref = json.load(f)
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(ours.keys()) == list(ref["keys"].keys())          # same keys, same ORDER
    assert ours == ref["keys"]
    assert sum(p.numel() for p in m.parameters()) == ref["n_params"]
    assert abs(float(m.logit_scale) - 2.6592600) < 1e-5            # log(1/0.07), used raw


def test_header_abi_and_library_agree():
    from eeg_image_decode_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "eegclip.h")).read()
    declared = set(re.findall(r"\b(?:int|long long|float|void\s*\*)\s*(eegclip_\w+)\s*\(", hdr))
    assert declared == set(_abi.PROTOTYPES), (declared ^ set(_abi.PROTOTYPES))
    libpath = os.path.join(ROOT, "eeg_image_decode_amd", "csrc", "libeegclip_hip.so")
    if not os.path.exists(libpath):
        from eeg_image_decode_amd import build
        build.build(verbose=False)
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = set(re.findall(r" T (eegclip_\w+)", out))
    assert declared <= exported, declared - exported
    lib = _abi.declare(ctypes.CDLL(libpath))                        # loads without a GPU; no compute call here
    assert lib.eegclip_abi_version() == _abi.ABI_VERSION
    # every header entry cites the reference lines it replaces
    assert hdr.count(".py:") >= 12


def test_no_cpu_fallback():
    from eeg_image_decode_amd._lib import EegclipError
    from eeg_image_decode_amd.atms import ATMS
    from eeg_image_decode_amd.loss import ClipLoss
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = ATMS()
    with pytest.raises(EegclipError):
        m(torch.zeros(2, 63, 250), torch.ones(2, dtype=torch.long))
    with pytest.raises(EegclipError):
        ClipLoss()(torch.zeros(4, 8), torch.zeros(4, 8), 1.0)
    with pytest.raises(EegclipError):
        m.encoder(torch.zeros(1))                                    # holders have no eager path


def test_product_never_imports_oracle_or_emulator():
    pkg = os.path.join(ROOT, "eeg_image_decode_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|hipemu|tests)\b", src, re.M), f
                assert "/root/reference" not in src, f
