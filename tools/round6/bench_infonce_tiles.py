"""round 6: the InfoNCE tile kernel's forward (tile launch + finalize launch = `logits_block`) and gradient pass by tile form at N = 2048 / 4096 / 8192, D = 1024,
one product (throughput mode) and three (parity mode, `parity_` rows); 256 x 256 tiles (tile code 255) against 128 x 128 (8 waves | 4 + 4 producer waves).   python tools/round6/bench_infonce_tiles.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib
from eeg_image_decode_amd.loss import split_planes

L = lib()
PEAK = 2500.0
st = torch.cuda.current_stream().cuda_stream


def ev_us(fn, reps=100, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
Dm = 1024
sc = torch.tensor([2.6593], device="cuda")
acc = torch.zeros(2, device="cuda")
out = {}
for N in (2048, 4096, 8192):
    a = torch.nn.functional.layer_norm(torch.randn(N, Dm, device="cuda", generator=g), (Dm,))
    b = torch.nn.functional.normalize(torch.randn(N, Dm, device="cuda", generator=g), dim=1)
    flop = 2.0 * N * N * Dm
    ws = int(L.eegclip_infonce_fused_workspace_floats(N, N))
    row = {}
    for planes in (1, 2):
        ap_, bp_ = split_planes(a, planes), split_planes(b, planes)
        buf = torch.empty(ws + 2 * N, device="cuda")
        G = torch.empty(N, N, device="cuda")
        arr = (_abi.InfonceProblem * 1)()
        arr[0] = _abi.InfonceProblem(q_hi=ap_[0].data_ptr(), q_lo=ap_[1].data_ptr() if planes == 2 else None, k_hi=bp_[0].data_ptr(),
                                     k_lo=bp_[1].data_ptr() if planes == 2 else None, col0=0, weight=0.5, part=buf.data_ptr(),
                                     diag=buf.data_ptr() + 4 * ws, lse=buf.data_ptr() + 4 * (ws + N), lse_k=None, G=G.data_ptr(), ldg=N)
        ref = None
        # waves: 2 = 8 waves, 3 = 4 MFMA + 4 producer waves, + 4 (bit 18) = the 8-wave form without the cross-barrier fragment prefetch; tile code 255 = 256 x 256 (one product only)
        forms = [("tile128_waves8", 128, 2), ("tile128_waves8_no_prefetch", 128, 6), ("tile128_waves4+4", 128, 3)]
        if planes == 1:
            forms += [("tile256_waves8", 255, 0), ("tile256_waves8_no_prefetch", 255, 4)]
        forms += [("auto", 0, 0)]
        for tag, tile, waves in forms:
            pl = planes | (tile << 8) | (waves << 16)
            acc.zero_()
            assert L.eegclip_infonce_fused_fwd(arr, 1, N, N, Dm, pl, N, sc.data_ptr(), acc.data_ptr(), st) == 0
            torch.cuda.synchronize()
            loss = float(acc[0])
            lse = buf[ws + N:ws + 2 * N].clone()
            if ref is None:
                ref = (loss, lse)
            dl, dlse = abs(loss - ref[0]), float((lse - ref[1]).abs().max())
            us = ev_us(lambda: L.eegclip_infonce_fused_fwd(arr, 1, N, N, Dm, pl, N, sc.data_ptr(), acc.data_ptr(), st))
            usg = ev_us(lambda: L.eegclip_infonce_fused_grad(arr, 1, N, N, Dm, pl, N, sc.data_ptr(), acc.data_ptr() + 4, st), reps=30)
            mult = 3 if planes == 2 else 1
            row[("parity_" if planes == 2 else "") + tag] = {"logits_block_us": round(us, 2), "TF": round(flop / us / 1e6, 1), "frac_of_bf16_peak": round(flop / us / 1e6 / PEAK, 4),
                                                             "mfma_work_frac_of_peak": round(mult * flop / us / 1e6 / PEAK, 4), "grad_us": round(usg, 2),
                                                             "loss_diff_to_first": dl, "lse_maxdiff_to_first": dlse}
    out[f"N{N}"] = row
    print(N, json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
