#!/usr/bin/env python3
"""Benchmark: EEG-CLIP contrastive TRAIN samples/s (global batch) -- BASELINE.json's metric on its configs[1]/[2].

One step = one iteration of the reference batch loop (Retrieval/ATMS_retrieval.py:209-250) on a synthetic batch that is
already resident in HBM: ATMS forward (train mode, dropout + batch-stat BatchNorm), image + text InfoNCE (0.99/0.01),
backward, AdamW, and the running train-accuracy GEMM + argmax vs 1654 class embeddings.  Nothing is skipped or cached.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: one process per GPU over RCCL; weak scaling (256 samples per GPU, global batch 256*N); ClipLoss all-gathers the
embeddings so every rank scores against the GLOBAL negatives, the flat gradient buffer is all-reduced once per step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_MFMA_TF = 157.3       # v_mfma_f32_*_f32 dense peak (f32 in / f32 acc)
N_CLASSES = 1654


def algorithmic_cost(name, desc, B):
    """(bound, work per launch) for the kernels that can dominate: flops for the MFMA GEMMs, HBM bytes for streaming ops.
    Per-unit figures (SURVEY.md section 8d / DESIGN.md): conv+pool fused = 75 taps x 40 filters x 36 outputs x 63 rows per
    sample; BN/ELU passes move the (B,40,63,36) fp32 tensor (362,880 B per sample per pass)."""
    y1 = B * 40 * 63 * 36 * 4
    if name == "eegclip_gemm_f32":
        return "mfma", 2.0 * desc.M * desc.N * desc.K, "flop"
    if name in ("eegclip_attention_fwd", "eegclip_attention_bwd"):
        per = 4 if name.endswith("fwd") else 10                    # QK^T + PV  |  + recompute, dP, dQ, dK, dV  (x 2 L^2 E flops)
        return "mfma", float(per * 64 * 64 * 62 * B * 4), "flop"
    tok = B * 63 * 250 * 4
    table = {                                                               # HBM-bound streaming over the (B,40,63,36) tensor y1
        "eegclip_tsconv_fwd": ("hbm", tok + y1),                            # read tokens, write y1
        "eegclip_tsconv_bwd_w": ("hbm", tok + y1),                          # read tokens + dy1
        "eegclip_tsconv_bwd_x": ("hbm", tok + y1),                          # read dy1, write token gradients
        "eegclip_sconv_fwd": ("hbm", y1),                                   # read y1 (z1 recomputed, y2 is tiny)
        "eegclip_sconv_bwd_w": ("hbm", y1),
        "eegclip_sconv_bwd_x_stats": ("hbm", y1),
        "eegclip_sconv_bwd_x_apply": ("hbm", 2 * y1),                       # read y1, write dy1
    }
    if name in table:
        return table[name][0], float(table[name][1]), "byte"
    return None


_KERNEL_OF = {"eegclip_attention_bwd": "eeg::attention_bwd_kernel<true>", "eegclip_attention_fwd": "eeg::attention_fwd_kernel<true>",
              "eegclip_tsconv_fwd": "eeg::tsconv_fwd_kernel", "eegclip_tsconv_bwd_w": "eeg::tsconv_bwd_w_kernel<7>",
              "eegclip_tsconv_bwd_x": "eeg::tsconv_bwd_x_kernel", "eegclip_sconv_fwd": "eeg::sconv_fwd_kernel",
              "eegclip_sconv_bwd_w": "eeg::sconv_bwd_w_kernel<128>", "eegclip_sconv_bwd_x_stats": "eeg::sconv_bwd_x_kernel<false>",
              "eegclip_sconv_bwd_x_apply": "eeg::sconv_bwd_x_kernel<true>"}


_GEMM_KERNEL = "eeg::gemm_f32_fast_kernel<true, true, true, false>"      # Y = X W^T launches (forward Linears); f02 is the largest of them


def pmc_traffic(op_name, B, desc=None):
    """HBM bytes per launch of the op's kernel from the committed rocprofv3 PMC summary (profiles/r1_pmc_hbm_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE passes of this very command, FETCH doubled per the gfx950 correction).  Only valid for the profiled B=256."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_hbm_traffic.json")
    if B != 256 or not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    if op_name == "eegclip_gemm_f32" and desc is not None and (desc.M, desc.N, desc.K) == (B * 64, 744, 250):
        d = table.get(_GEMM_KERNEL)                       # the QKV projection is the largest launch of this instantiation
        return round(d["hbm_bytes_largest_launch"]) if d and "hbm_bytes_largest_launch" in d else None
    if op_name not in _KERNEL_OF:
        return None
    d = table.get(_KERNEL_OF[op_name])
    return round(d["hbm_bytes_per_launch"]) if d and "hbm_bytes_per_launch" in d else None


def build(world, rank, B, seed=0):
    from eeg_image_decode_amd import dist as edist
    from eeg_image_decode_amd import optim, synthetic as syn
    from eeg_image_decode_amd.atms import ATMS
    torch.manual_seed(seed)                                     # identical random init on every rank
    model = ATMS().cuda().train()
    edist.configure_loss_for_world(model.loss_func, rank, world)
    opt = optim.AdamW(model.parameters(), lr=3e-4)
    pool = []
    classes = torch.from_numpy(syn.unit_features(seed + 1, N_CLASSES, tag="classes")).cuda()
    for i in range(4):                                           # a small pool of distinct resident batches
        s = seed + 100 + 17 * rank + i
        labels = torch.from_numpy(np.random.default_rng(s).integers(0, N_CLASSES, B)).cuda()
        pool.append(dict(eeg=torch.from_numpy(syn.eeg_batch(s, B)).cuda(), img=classes[labels].contiguous(),
                         txt=torch.from_numpy(syn.unit_features(s, B, tag="txt")).cuda(), labels=labels))
    return model, opt, pool, classes


def cpu_baseline(B, seconds=12.0):
    """The oracle ("port" of the reference's PyTorch-CPU step) timed on this box's host cores: same step, same shapes."""
    from eeg_image_decode_amd import synthetic as syn
    from oracle import atms as oatms, loops as oloops, loss as oloss
    ncpu = os.cpu_count() or 1
    state = oloops.torch_state(syn.make_state(0, oatms.state_spec()))
    tr = oloops.OracleTrainer(state)
    x = torch.from_numpy(syn.eeg_batch(1, B))
    img = torch.from_numpy(syn.unit_features(1, B, tag="img"))
    txt = torch.from_numpy(syn.unit_features(1, B, tag="txt"))
    classes = torch.from_numpy(syn.unit_features(2, N_CLASSES, tag="classes"))
    ids = torch.full((B,), 1, dtype=torch.long)

    def step():
        _, z = tr.step(x, ids, img, txt)
        oloss.train_accuracy_predictions(z, classes, tr.P["logit_scale"])

    # torch-CPU oversubscribes badly on many-core hosts for these small ops: probe a few thread counts (one step each) and
    # keep the fastest -- the baseline is the reference path at its best on this box, with the thread count reported.
    step()                                                       # warm-up (first step pays allocator / MKL-DNN setup)
    best = None
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        step()
        t = time.perf_counter()
        step()
        t = time.perf_counter() - t
        if best is None or t < best[0]:
            best = (t, nt)
    torch.set_num_threads(best[1])
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds or n >= 40:
            break
    return {"value": round(n * B / dt, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} full train steps at B={B} (fwd, img+txt InfoNCE, bwd, AdamW, accuracy GEMM), fp32, torch-CPU oracle, "
                      f"{dt:.1f}s after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU (BASELINE configs[1]: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="time every kernel of the fwd/bwd plans with HIP events and print a table")
    ap.add_argument("--roofline-kernel", default="auto", help="op name for the roofline object (auto = the op with the largest total time)")
    args = ap.parse_args()

    from eeg_image_decode_amd import dist as edist
    from eeg_image_decode_amd import retrieval
    rank, local_rank, world = edist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    B = args.batch
    model, opt, pool, classes = build(world, rank, B)
    loss_acc = torch.zeros((), device="cuda")
    correct = torch.zeros(1, dtype=torch.int32, device="cuda")

    def step(i):
        d = pool[i % len(pool)]
        retrieval.contrastive_step(model, opt, d["eeg"], 1, d["img"], d["txt"], d["labels"], classes, loss_acc, correct)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # plan building + clock ramp (untimed, before the official warm-up): the first steps compile nothing but build the launch plans,
    # and a cold GPU needs a few hundred milliseconds of load to reach its sustained clocks
    t_ramp = time.perf_counter()
    i_ramp = 0
    while time.perf_counter() - t_ramp < 0.4 or i_ramp < 3:
        step(i_ramp)
        i_ramp += 1
        if i_ramp % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    eng = model._engine()
    plans = {k: v for k, v in eng.plans.items()}
    # ---- choose the kernels to time live (HIP events on the launch stream, inside the timed region) -----------------
    timed_sel = {}
    if args.breakdown:
        for k, pl in plans.items():
            pl.use_side_stream = False          # one stream: per-kernel times that add up (the headline run overlaps the weight-gradient kernels)
            pl.time_ops(range(len(pl.ops)))
    else:
        # pass 0 (untimed by the headline clock): 3 instrumented single-stream steps to find the dominant kernel, unless one is named
        for k, pl in plans.items():
            pl.use_side_stream = False
            pl.time_ops(range(len(pl.ops)))
        for i in range(3):
            step(i)
        for k, pl in plans.items():
            pl.use_side_stream = True
        tot = {}
        for k, pl in plans.items():
            for idx, v in pl.timings_ms().items():
                tot[(k, idx)] = float(np.mean(v))
            pl.time_ops([])
        cands = sorted(tot.items(), key=lambda kv: -kv[1])
        for (k, idx), ms in cands:
            name = plans[k].ops[idx][2]
            # auto: the largest launch whose LIVE duration is its own -- plans with side-stream ops (the backward) overlap kernels of two
            # streams inside the timed region, which stretches every per-launch duration there; name an op explicitly to time one anyway
            if args.roofline_kernel == "auto" and any(op[3] for op in plans[k].ops):
                continue
            if args.roofline_kernel in ("auto", name) and algorithmic_cost(name, _desc_of(plans[k], idx), B) is not None:
                timed_sel = {(k, idx): name}
                plans[k].time_ops([idx])
                break

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_enq = time.perf_counter() - t0          # host time to ENQUEUE the steps (no sync inside a step): must stay below the GPU time
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)
    ms_per_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    final_loss = float(loss_acc) / (i_ramp + args.warmup + args.steps + (0 if args.breakdown else 3))

    roof = None
    if args.breakdown:
        rows = []
        for k, pl in plans.items():
            for idx, v in pl.timings_ms().items():
                name = pl.ops[idx][2]
                d = _desc_of(pl, idx)
                tag = f"{name}[{d.M}x{d.N}x{d.K}{'/sk' + str(d.split_k) if d.split_k > 1 else ''}]" if d is not None else name
                rows.append((float(np.mean(v[-args.steps:])), k[0], idx, tag))
        rows.sort(reverse=True)
        if rank == 0:
            tot = sum(r[0] for r in rows)
            print(f"# per-kernel HIP-event breakdown (ms/step, mean over {args.steps} steps); sum of kernels = {tot:.3f} ms, step = {ms_per_step:.3f} ms", file=sys.stderr)
            for ms, ph, idx, tag in rows:
                print(f"#  {ms:8.4f} ms  {100 * ms / tot:5.1f}%  {ph}{idx:02d}  {tag}", file=sys.stderr)
    else:
        for (k, idx), name in timed_sel.items():
            v = plans[k].timings_ms()[idx][-args.steps:]
            ms = float(np.mean(v))
            bound, work, unit = algorithmic_cost(name, _desc_of(plans[k], idx), B)
            if bound == "mfma":
                ach, peak, u = work / (ms * 1e-3) / 1e12, PEAK_F32_MFMA_TF, "TFLOP/s"
            else:
                ach, peak, u = work / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            d = _desc_of(plans[k], idx)
            roof = {"kernel": name + (f"[{d.M}x{d.N}x{d.K}]" if d is not None else ""), "bound": bound, "achieved": round(ach, 2),
                    "peak": peak, "unit": u, "frac": round(ach / peak, 4), "traffic": pmc_traffic(name, B, d), "avg_launch_ms": round(ms, 5),
                    "algorithmic_work_per_launch": work, "work_unit": unit}

    out = {
        "metric": "EEG-CLIP contrastive train samples/sec (global batch)", "value": round(value, 1), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: full ATM-S EEG encoder (63ch x 250t -> 1024-d) contrastive train step vs frozen 1024-d CLIP "
                               "embeddings, 256 samples/GPU" + (f", global batch {world * B} with RCCL all-gather negatives (configs[2])" if world > 1 else ""),
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}", "n_classes_for_accuracy": N_CLASSES,
                   "optimizer": "AdamW lr 3e-4 (fused)", "final_mean_loss": round(final_loss, 4),
                   "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 4)},
        "roofline": roof,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def _desc_of(plan, idx):
    fn, a, name = plan.ops[idx][:3]
    if name == "eegclip_gemm_f32":
        return a[0]._obj
    return None


if __name__ == "__main__":
    main()
