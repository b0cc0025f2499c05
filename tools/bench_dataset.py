"""Input-pipeline measurements on one MI355X: (1) eegclip_stage_eeg on a real-shaped chunk -- float64 (1000 images, 4 reps, 63, 250)
-> float32, HIP events, against the HBM roofline (algorithmic bytes = 8 read + 4 written per sample value); (2) construction of an EEGDataset from
the on-disk tree (pickle load + H2D + staging); (3) shuffled batches of 256 from the resident split (4 gather launches per batch).
Prints one JSON object."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd import synthetic as syn
from eeg_image_decode_amd._lib import check, lib
from eeg_image_decode_amd.datasets import EEGDataset

HBM_PEAK_GBS = 8000.0


def main():
    out = {}
    n, reps, C, T = 1000, 4, 63, 250
    src = torch.randn(n, reps, C, T, dtype=torch.float64, device="cuda")
    dst = torch.empty(n * reps, C, T, dtype=torch.float32, device="cuda")
    tidx = torch.arange(T, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for mean in (0, 1):
        run = lambda: check(lib().eegclip_stage_eeg(src.data_ptr(), dst.data_ptr(), n, reps, C, T, None if not mean else tidx.data_ptr(), T, mean, st), "stage")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nbytes = src.numel() * 8 + (dst.numel() if not mean else dst.numel() // reps) * 4
        out["stage_eeg_mean_over_reps" if mean else "stage_eeg_cast_window"] = {"us": round(us, 1), "GBs": round(nbytes / us / 1e3, 1),
                                                                                "frac_of_hbm_peak": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 3)}
    root = tempfile.mkdtemp(prefix="things_bench_")
    try:
        cfg = syn.write_things_eeg_tree(root, 3, subjects=("sub-01",), channels=63, n_times=300, dt=0.004, train_classes=100, test_classes=200, test_reps=4)
        t0 = time.perf_counter()
        ds = EEGDataset(cfg["data_path"], subjects=["sub-01"], train=True, config=cfg, features_dir=root)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fbytes = os.path.getsize(os.path.join(cfg["data_path"], "sub-01", "preprocessed_eeg_training.npy"))
        out["dataset_construct"] = {"file_MB": round(fbytes / 1e6, 1), "seconds": round(dt, 3), "samples": len(ds), "note": "pickle load + H2D + staging"}
        ld = ds.loader(batch_size=256, shuffle=True, drop_last=True)
        for _ in ld:
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = 0
        for _ in range(5):
            for b in ld:
                nb += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["device_loader_b256"] = {"batches": nb, "us_per_batch": round(dt / nb * 1e6, 1), "samples_per_s": round(nb * 256 / dt, 1)}
        # the feature dump that feeds the diffusion prior (Generation notebooks: eval-mode embeddings of the whole training split)
        from eeg_image_decode_amd import retrieval
        from eeg_image_decode_amd.atms import ATMS
        m = ATMS().cuda()
        retrieval.get_eegfeatures("sub-01", m, ds.loader(batch_size=1000), "cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            feats = retrieval.get_eegfeatures("sub-01", m, ds.loader(batch_size=1000), "cuda")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["feature_dump_b1000"] = {"samples": len(ds), "ms_per_pass": round(dt / 5 * 1e3, 2), "samples_per_s": round(5 * len(ds) / dt, 1),
                                     "whole_subject_66160_s": round(66160 / (5 * len(ds) / dt), 3)}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
