"""Fused InfoNCE kernels (csrc/infonce_fused.hip) against the round-1 logits kernel (csrc/logits_bf16.hip), HIP-event timed back to back; run it
under `rocprofv3 --kernel-trace --stats` for the per-kernel durations.   python tools/bench_infonce_fused.py [--out file.json]"""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib
from eeg_image_decode_amd.loss import split_planes

L = lib()
PEAK = 2500.0


def ev_us(fn, reps=200, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def problem(qp, kp, n, N, planes, ws, col0=0, G=None, lse_k=None):
    buf = torch.empty(ws + 2 * n, device="cuda")
    p = _abi.InfonceProblem(q_hi=qp[0].data_ptr(), q_lo=qp[1].data_ptr() if planes == 2 else None, k_hi=kp[0].data_ptr(),
                            k_lo=kp[1].data_ptr() if planes == 2 else None, col0=col0, weight=0.5, part=buf.data_ptr(), diag=buf.data_ptr() + 4 * ws,
                            lse=buf.data_ptr() + 4 * (ws + n), lse_k=lse_k, G=G.data_ptr() if G is not None else None, ldg=N)
    return p, buf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(0)
    Dm = 1024
    sc = torch.tensor([2.6593], device="cuda")
    acc = torch.zeros(2, device="cuda")
    out = {}
    for n, N, label in ((2048, 2048, "square_2048"), (256, 2048, "rank_block_256x2048"), (256, 256, "square_256")):
        a = torch.nn.functional.layer_norm(torch.randn(N, Dm, device="cuda", generator=g), (Dm,))
        b = torch.nn.functional.normalize(torch.randn(N, Dm, device="cuda", generator=g), dim=1)
        flop = 2.0 * n * N * Dm
        ws = int(L.eegclip_infonce_fused_workspace_floats(n, N))
        row = {}
        if n == N and n % 128 == 0:
            a16, b16 = split_planes(a, 1)[0], split_planes(b, 1)[0]
            C = torch.empty(n, N, device="cuda")
            us = ev_us(lambda: L.eegclip_logits_bf16(a16.data_ptr(), b16.data_ptr(), C.data_ptr(), n, N, Dm, N, sc.data_ptr(), st))
            row["r1_logits_kernel_with_stores"] = {"us": round(us, 2), "TF": round(flop / us / 1e6, 1), "frac": round(flop / us / 1e6 / PEAK, 4)}
        for planes in (1, 2):
            ap_, bp_ = split_planes(a, planes), split_planes(b, planes)
            qp = (ap_[0][:n], ap_[1][:n] if planes == 2 else None)
            for tile in (0, 64, 128):
                if tile == 128 and (n % 128 or N % 128):
                    continue
                keep = []
                for nblk in (1, 2):
                    arr = (_abi.InfonceProblem * nblk)()
                    for i in range(nblk):
                        arr[i], buf = problem(qp, bp_, n, N, planes, ws)
                        keep.append(buf)
                    for waves in ((1, 2, 3) if tile == 128 else (1, 3) if tile == 64 else (0,)):          # 4 waves (one per SIMD) | 8 waves (two per SIMD; 128-tiles) | 4 MFMA + 4 producer waves (the default)
                        pl = planes | (tile << 8) | (waves << 16)
                        assert L.eegclip_infonce_fused_fwd(arr, nblk, n, N, Dm, pl, n, sc.data_ptr(), acc.data_ptr(), st) == 0
                        us = ev_us(lambda: L.eegclip_infonce_fused_fwd(arr, nblk, n, N, Dm, pl, n, sc.data_ptr(), acc.data_ptr(), st))
                        tag = f"fwd_planes{planes}_tile{tile or 'auto'}{'_' + {1: 'waves4', 2: 'waves8', 3: 'waves4+4producers'}[waves] if waves else ''}_blocks{nblk}"
                        row[tag] = {"us": round(us, 2), "TF_algorithmic": round(nblk * flop / us / 1e6, 1), "frac": round(nblk * flop / us / 1e6 / PEAK, 4)}
            G = torch.empty(n, N, device="cuda")
            arr = (_abi.InfonceProblem * 1)()
            arr[0], buf = problem(qp, bp_, n, N, planes, ws, G=G)
            L.eegclip_infonce_fused_fwd(arr, 1, n, N, Dm, planes, n, sc.data_ptr(), acc.data_ptr(), st)
            us = ev_us(lambda: L.eegclip_infonce_fused_grad(arr, 1, n, N, Dm, planes, n, sc.data_ptr(), acc.data_ptr() + 4, st))
            row[f"grad_planes{planes}"] = {"us": round(us, 2), "TF_algorithmic": round(flop / us / 1e6, 1)}
        us = ev_us(lambda: L.eegclip_split_bf16(a.data_ptr(), ap_[0].data_ptr(), ap_[1].data_ptr(), a.numel(), st))
        row["split_one_operand_hi_lo"] = {"us": round(us, 2), "GBs": round(a.numel() * 8 / us / 1e3, 1)}
        out[label] = row
        print(label, json.dumps(row, indent=0).replace("\n", " "), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
