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
PEAK_BF16_MFMA_TF = 2500.0     # v_mfma_f32_*_bf16 dense peak
N_CLASSES = 1654
PREC_BF16X3 = 1


# algorithmic HBM bytes per sample of the fused transformer-block launches (csrc/token_block.hip): what has to cross HBM given that the
# weight-gradient GEMMs and the attention backward read their operands from it
_ROW, _ROWF, _QKV, _CTX = 64 * 250 * 4, 64 * 256 * 4, 64 * 744 * 4, 64 * 248 * 4
# (_ROWF = 64 x 256 x 4 B is also one sample of a token-plane tensor: [hi | lo][64][256] bf16)
TOKEN_BLOCK_BYTES = {
    "fwd": 63 * 250 * 4 + 4 * _ROW + _QKV + 6 * _ROWF,                 # x in; h, r1, r2, n3 + qkv + f1 (fp32) + x, h, ctx, n1, g1 (token planes) out
    0: 3 * _ROW + _ROWF + _ROW + 3 * _ROWF + _CTX,                     # dn3, r2, r1, f1 in; dr1, df2 / dg1 / da1 (token planes), dctx out
    1: 3 * _ROWF + 2 * _ROW + _ROWF,                                   # dq / dk / dv planes, dr1 in; dr1 + its planes out
}


def algorithmic_cost(name, desc, B):
    """(bound, work per launch) for the kernels that can dominate: flops for the MFMA GEMMs, HBM bytes for streaming ops.
    Per-unit figures (SURVEY.md section 8d / DESIGN.md): conv+pool fused = 75 taps x 40 filters x 36 outputs x 63 rows per
    sample; BN/ELU passes move the (B,40,63,36) fp32 tensor (362,880 B per sample per pass)."""
    y1 = B * 40 * 63 * 36 * 4
    if name in ("eegclip_gemm_f32", "eegclip_head_gemm"):
        return "mfma", 2.0 * desc.M * desc.N * desc.K, "flop"
    if name == "eegclip_wgrad_tok":                                # the block's weight gradients from token planes: sum of 2 M N K over the launch's problems
        return "mfma", float(desc.flops), "flop"
    if name == "eegclip_wgrad_tok_reduce":                         # its ordered slab reduction: time counted in the family, no additional algorithmic work
        return "mfma", 0.0, "flop"
    if name in ("eegclip_attention_fwd", "eegclip_attention_bwd", "eegclip_attention_bwd_x3"):
        per = 4 if name.endswith("fwd") else 10                    # QK^T + PV  |  + recompute, dP, dQ, dK, dV  (x 2 L^2 E flops)
        return "mfma", float(per * 64 * 64 * 62 * B * 4), "flop"
    if name == "eegclip_token_block_fwd":
        return "hbm", float(B * TOKEN_BLOCK_BYTES["fwd"]), "byte"
    if name == "eegclip_token_block_bwd":
        return ("hbm", float(B * TOKEN_BLOCK_BYTES[desc]), "byte") if desc in TOKEN_BLOCK_BYTES else None
    tok = B * 63 * 250 * 4
    # the conv stack recomputed from the token rows (csrc/cstack*.hip, round 5): contractions on the bf16 matrix cores, HBM traffic = the token rows.
    # algorithmic flops per sample: tap contraction 2 x 40 x 25 x 63 x 36, spatial contraction (and its transposes) 2 x 40 x 40 x 63 x 36
    tap, spat = 2.0 * 40 * 25 * 63 * 36, 2.0 * 40 * 40 * 63 * 36
    cstack = {"eegclip_cstack_stats1": tap, "eegclip_cstack_fwd": tap + spat, "eegclip_cstack_bwd_stats": tap + spat,
              "eegclip_cstack_bwd_apply": 3 * tap + spat,        # y1 + dz1 + E = dy1^T taps + the taps gradient
              "eegclip_cstack_bwd_w2": tap + spat, "eegclip_cstack_pack": 0.0, "eegclip_cstack_pack_t": 0.0, "eegclip_bn_finalize_rows": 0.0}
    if name in cstack:
        return "mfma", B * cstack[name], "flop"
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


_KERNEL_OF = {"eegclip_attention_bwd": "eeg::attention_bwd_kernel<true>", "eegclip_attention_bwd_x3": "eeg::attention_bwd_x3_kernel<true>", "eegclip_attention_fwd": "eeg::attention_fwd_kernel<true>",
              "eegclip_tsconv_fwd": "eeg::tsconv_fwd_kernel", "eegclip_tsconv_bwd_w": "eeg::tsconv_bwd_w_kernel<7>",
              "eegclip_tsconv_bwd_x": "eeg::tsconv_bwd_x_kernel", "eegclip_sconv_fwd": "eeg::sconv_fwd_kernel",
              "eegclip_sconv_bwd_w": "eeg::sconv_bwd_w_x3_kernel<128>", "eegclip_sconv_bwd_x_stats": "eeg::sconv_bwd_x_kernel<false, true>",
              "eegclip_sconv_bwd_x_apply": "eeg::sconv_bwd_x_kernel<true, true>"}


PMC_SUMMARY = os.path.join("profiles", "r6_pmc_hbm_traffic.json")


def pmc_traffic(family, B):
    """(HBM bytes per launch averaged over the family's launches of one step, source) from the COMMITTED rocprofv3 PMC summary of this very
    command (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 correction; tools/pmc_summary.py): PMC counters cannot
    be collected from inside the process, so this is a stored number next to the live ones -- `traffic_source` says so.  Only for the profiled
    batch size; (None, None) otherwise."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        table = json.load(f)
    if table.get("batch") != B:
        return None, None
    d = table.get("families", {}).get(family)
    return (round(d["hbm_bytes_per_launch"]), PMC_SUMMARY) if d and "hbm_bytes_per_launch" in d else (None, None)


def pmc_mfma_busy(family, B):
    """matrix-pipe busy fraction of a family from the same committed PMC summary (None if absent)"""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    if table.get("batch") != B:
        return None
    return table.get("families", {}).get(family, {}).get("mfma_busy_frac")


def algorithmic_bytes(name, desc, B):
    """HBM bytes a launch has to move at least (operands once in, results once out): the traffic the PMC figure is compared with"""
    if name in ("eegclip_gemm_f32", "eegclip_head_gemm"):             # (plane operands are 4 bytes per element too: hi | lo)
        return 4.0 * (desc.M * desc.K + desc.K * desc.N + desc.M * desc.N)
    if name == "eegclip_wgrad_tok":                                # token planes are 4 bytes per element (hi | lo); partial tiles are not algorithmic
        return float(desc.bytes)
    if name == "eegclip_wgrad_tok_reduce":
        return 0.0
    if name in ("eegclip_attention_bwd_x3", "eegclip_attention_bwd"):
        return float(B * 64 * (2 * 744 + 248) * 4)                 # qkv + dctx in, dqkv out
    if name.startswith("eegclip_cstack_") or name == "eegclip_bn_finalize_rows":
        tok = B * 63 * 250 * 4.0                                    # the token rows in (apply: + their gradients out); packs / finalize: nothing per sample
        return {"eegclip_cstack_bwd_apply": 2 * tok, "eegclip_cstack_pack": 0.0, "eegclip_cstack_pack_t": 0.0, "eegclip_bn_finalize_rows": 0.0}.get(name, tok)
    c = algorithmic_cost(name, desc, B)
    return c[1] if c and c[2] == "byte" else None


# algorithmic flops per sample of the fused transformer-block launches (2 M N K of every Linear + 4 L^2 E per head of the attention forward; the
# backward parts hold the dX GEMMs only -- the weight gradients are the wgrad_tok launches): the MFMA-roof sibling of the family's HBM fraction
TOKEN_BLOCK_FLOPS = {"fwd": 2.0 * 63 * 250 * 250 + 2.0 * 64 * 250 * 744 + 4 * 4.0 * 64 * 64 * 62 + 2.0 * 64 * 248 * 250 + 2 * 2.0 * 64 * 250 * 256,
                     0: 2 * 2.0 * 64 * 250 * 256 + 2.0 * 64 * 248 * 250, 1: 2.0 * 64 * 744 * 250}


def algorithmic_flops(name, desc, B):
    if name == "eegclip_token_block_fwd":
        return B * TOKEN_BLOCK_FLOPS["fwd"]
    if name == "eegclip_token_block_bwd":
        return B * TOKEN_BLOCK_FLOPS.get(desc, 0.0)
    c = algorithmic_cost(name, desc, B)
    return c[1] if c and c[2] == "flop" else None


TIME_EVERY = 8


def family_of(name, desc):
    """kernels are grouped for the roofline: every GEMM launch of one arithmetic (they are one kernel template), the fused transformer-block
    launches (forward + the two backward parts), the attention kernels, each other op on its own"""
    if name == "eegclip_gemm_f32":
        return "gemm_bf16x3" if (desc.precision & 0xff) == PREC_BF16X3 else "gemm_f32"
    if name.startswith("eegclip_wgrad_tok") or name == "eegclip_head_gemm":
        return "gemm_bf16x3"
    if name.startswith("eegclip_token_block_"):
        return "token_block"
    if name.startswith("eegclip_cstack_") or name == "eegclip_bn_finalize_rows":
        return "conv_stack"
    if name == "eegclip_attention_bwd_x3":
        return "attention_bf16x3"
    if name.startswith("eegclip_attention_"):
        return "attention_f32_mfma"
    return name


def family_peak(fam, bound):
    """(peak, scale from work / ms to the unit, unit)"""
    if bound == "mfma":
        # bf16x3: three bf16 MFMA products per algorithmic multiply-add -> the pipe's ceiling for ALGORITHMIC flops is a third of 2.5 PF
        return (PEAK_BF16_MFMA_TF / 3.0 if fam in ("gemm_bf16x3", "attention_bf16x3", "conv_stack") else PEAK_F32_MFMA_TF), 1e-3 * 1e12, "TFLOP/s"
    return PEAK_HBM_GBS, 1e-3 * 1e9, "GB/s"


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


def _ev_ms(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def secondary():
    """The other rows of SURVEY.md section 8, timed by this process so that they are driver-run numbers, not builder-run ones: InfoNCE at global
    batch 2048 (north_star's kernel target), diffusion-prior training at batch 1024 (configs[3]), the 50-step prior sampling chain, the SDXL
    cross-attention kernel at 8 images x CFG (configs[4]), the SDXL-shaped sampling loop, and one end-to-end pass dataset files -> loader -> train_model."""
    out = {}

    def section(name, fn):
        try:
            t0 = time.perf_counter()
            out[name] = fn()
            out[name]["bench_wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:                                # a secondary line must never take the headline down with it
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    section("infonce_global_batch_2048", _sec_infonce)
    section("infonce_global_batch_4096", lambda: _sec_infonce(4096, light=True))       # where does the tile shape reach north_star's 0.40 of the bf16 roof?
    section("infonce_global_batch_8192", lambda: _sec_infonce(8192, light=True))
    section("infonce_per_rank_block", _sec_infonce_per_rank)
    section("exact_fp32_products", _sec_exact_fp32)
    section("bench_joint", _sec_joint)
    section("prior_train_batch_1024", _sec_prior_train)
    section("prior_sampling_chain", _sec_prior_chain)
    section("sdxl_cross_attention", _sec_cross_attn)
    section("sdxl_sampling_loop", _sec_sdxl_loop)
    section("vae_decode_1024px", _sec_vae_decode)
    section("end_to_end_dataset_loader_train_model", _sec_end_to_end)
    return out


def _sec_vae_decode():
    """the VAE decode that ends a sampling loop (Generation/custom_pipeline.py:421) at 1024 x 1024 on csrc/vae.hip: SDXL's VAE layout (83.65 M parameters), random
    weights -- diffusers / the checkpoint are absent offline (stand_in)"""
    from eeg_image_decode_amd import vae
    r = vae.bench_decode(images=1, latent=128)
    r["frac_of_bf16_mfma_peak"] = round(r["algorithmic_TFLOPs"] / PEAK_BF16_MFMA_TF, 4)
    r["workload"] = "AutoencoderKL decode of one 128 x 128 x 4 latent -> 1024 x 1024 x 3, bf16 activations, fp32 accumulation"
    return r


def _sec_infonce(N=2048, Dm=1024, light=False):
    """north_star's kernel target: the InfoNCE logits at global batch 2048 on the bf16 matrix cores.  `logits_block` = ONE fused launch over one
    N x N block (tile kernel + the small finalize kernel): 2 N^2 D algorithmic flops, the logits never leave the chip.  `clip_loss_*` = the whole
    ClipLoss call (feature split, both blocks of the symmetric loss = 4 N^2 D flops, and for forward_backward the gradient matrix + dA GEMM)."""
    import ctypes
    from eeg_image_decode_amd import _abi
    from eeg_image_decode_amd._lib import lib
    from eeg_image_decode_amd.loss import ClipLoss, split_planes
    L = lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.nn.functional.layer_norm(torch.randn(N, Dm, device="cuda", generator=g), (Dm,))        # EEG embeddings leave a LayerNorm
    b = torch.nn.functional.normalize(torch.randn(N, Dm, device="cuda", generator=g), dim=1)          # CLIP targets are unit norm
    sc = torch.tensor(2.6593, device="cuda")
    flop = 2.0 * N * N * Dm
    res = {"workload": f"CLIP-symmetric InfoNCE, N = {N} ({N // 256} x 256 gathered), D = {Dm}; fractions are of the dense bf16 MFMA peak (2.5 PFLOP/s)"}
    st = torch.cuda.current_stream().cuda_stream
    ws = int(L.eegclip_infonce_fused_workspace_floats(N, N))
    ref = None
    for mode, planes in (("f32", 2), ("bf16", 1)):
        ap, bp = split_planes(a, planes), split_planes(b, planes)
        buf = torch.empty(ws + 2 * N, device="cuda")
        acc = torch.zeros(2, device="cuda")
        pr = (_abi.InfonceProblem * 1)(_abi.InfonceProblem(q_hi=ap[0].data_ptr(), q_lo=ap[1].data_ptr() if planes == 2 else None, k_hi=bp[0].data_ptr(),
                                                           k_lo=bp[1].data_ptr() if planes == 2 else None, col0=0, weight=0.5, part=buf.data_ptr(),
                                                           diag=buf.data_ptr() + 4 * ws, lse=buf.data_ptr() + 4 * (ws + N), lse_k=None, G=None, ldg=0))
        ms_blk = _ev_ms(lambda: L.eegclip_infonce_fused_fwd(pr, 1, N, N, Dm, planes, N, sc.data_ptr(), acc.data_ptr(), st), 100, warm=10)
        mult = 3.0 if planes == 2 else 1.0                 # MFMA products per algorithmic multiply-add
        row = {
            "arithmetic": "bf16x3 split products (logits within ~5e-5 of fp32)" if planes == 2 else "one bf16 product (features rounded to bf16)",
            # which mode meets north_star's 1e-3 logit tolerance: the split products do (5e-5); one bf16 product does not (~3e-2 at |logit| ~ 16)
            "meets_north_star_logit_tolerance_1e-3": planes == 2,
            "logits_block_us": round(ms_blk * 1e3, 2), "logits_block_algorithmic_TFLOPs": round(flop / ms_blk / 1e9, 1),
            "logits_block_frac_of_bf16_mfma_peak": round(flop / ms_blk / 1e9 / PEAK_BF16_MFMA_TF, 4),
            "logits_block_mfma_work_frac_of_peak": round(mult * flop / ms_blk / 1e9 / PEAK_BF16_MFMA_TF, 4)}
        if not light:
            lf = ClipLoss(logits_dtype=mode)
            with torch.no_grad():
                ms_f = _ev_ms(lambda: lf(a, b, sc), 30)
                loss = float(lf(a, b, sc))
            ar = a.clone().requires_grad_()
            ms_fb = _ev_ms(lambda: lf(ar, b, sc), 20)            # ClipLoss computes the gradients in its forward (one pass)
            ref = loss if ref is None else ref
            row.update({"clip_loss_forward_us": round(ms_f * 1e3, 1), "clip_loss_forward_backward_us": round(ms_fb * 1e3, 1),
                        "loss": round(loss, 6), "abs_loss_difference_to_parity_mode": round(abs(loss - ref), 7)})
        res["parity_mode" if mode == "f32" else "throughput_mode"] = row
    return res


def _sec_infonce_per_rank(W=8, n=256, Dm=1024, rank=3):
    """configs[2] as ONE GPU sees it: local_loss + gather_with_grad, world 8 x 256 rows -- per target two row-sharded blocks (A_r, B_all) / (B_r, A_all)
    of 256 x 2048 x 1024 (positives at column 256 r), image + text targets in one step.  `blocks_*` = the fused launches alone over the 4 blocks
    (logits tiles + finalize, then the gradient tiles); `sharded_*` = everything a rank computes between the all-gather and the reduce-scatter
    (plane splits of the gathered matrices, the blocks, the dA / dA_all GEMMs)."""
    from eeg_image_decode_amd import loss as ploss
    g = torch.Generator(device="cuda").manual_seed(1)
    N = W * n
    a_all = torch.nn.functional.layer_norm(torch.randn(N, Dm, device="cuda", generator=g), (Dm,))
    b_alls = [torch.nn.functional.normalize(torch.randn(N, Dm, device="cuda", generator=g), dim=1) for _ in range(2)]
    sl = slice(rank * n, (rank + 1) * n)
    a_, bs = a_all[sl].contiguous(), [b[sl].contiguous() for b in b_alls]
    sc = torch.tensor([2.6593], device="cuda")
    res = {"workload": f"row-sharded InfoNCE of one rank at world {W}: 2 targets x 2 blocks of {n} x {N} x {Dm}; tile kernel grid = 4 blocks x "
                       f"{(n // 64) * (N // 64)} workgroups of 64 x 64 logits (256 CUs)"}
    flop = 4 * 2.0 * n * N * Dm
    for mode, planes in (("parity_mode", 2), ("throughput_mode", 1)):
        ap_all = ploss.split_planes(a_all, planes)
        bps = [ploss.split_planes(b, planes) for b in b_alls]
        cut = lambda p: (p[0][sl], p[1][sl] if planes == 2 else None)
        blocks = []
        for t, w in enumerate((0.99, 0.01)):
            blocks += [(cut(ap_all), bps[t], rank * n, 0.5 * w), (cut(bps[t]), ap_all, rank * n, 0.5 * w)]
        acc = torch.zeros(2, device="cuda")
        ms_f = _ev_ms(lambda: ploss.fused_infonce(blocks, n, N, Dm, planes, n, sc, acc, []), 50, warm=5)
        ms_fb = _ev_ms(lambda: ploss.fused_infonce(blocks, n, N, Dm, planes, n, sc, acc, [(i, None) for i in range(4)]), 50, warm=5)
        ms_sh = _ev_ms(lambda: ploss.sharded_blocks(True, True, rank, W, a_, bs, a_all, b_alls, (0.99, 0.01), sc, acc, True, [False, False], True, planes), 30, warm=3)
        mult = 3.0 if planes == 2 else 1.0
        res[mode] = {"blocks_forward_us": round(1e3 * ms_f, 1), "blocks_forward_backward_us": round(1e3 * ms_fb, 1),
                     "blocks_forward_algorithmic_TFLOPs": round(flop / ms_f / 1e9, 1),
                     "blocks_forward_frac_of_bf16_mfma_peak": round(flop / ms_f / 1e9 / PEAK_BF16_MFMA_TF, 4),
                     "blocks_forward_mfma_work_frac_of_peak": round(mult * flop / ms_f / 1e9 / PEAK_BF16_MFMA_TF, 4),
                     "sharded_loss_forward_backward_us": round(1e3 * ms_sh, 1)}
    return res


def _sec_exact_fp32(B=256, steps=20):
    """the headline step with EXACT fp32 products in every Linear (EEGCLIP_GEMM_PRECISION=f32: v_mfma_f32_16x16x4_f32, launch-per-Linear plans) --
    the strict-precision reading of BASELINE configs[1] next to the default split-bf16 arithmetic"""
    from eeg_image_decode_amd import retrieval
    old = os.environ.get("EEGCLIP_GEMM_PRECISION")
    os.environ["EEGCLIP_GEMM_PRECISION"] = "f32"
    try:
        model, opt, pool, classes = build(1, 0, B)
        loss_acc, correct = torch.zeros((), device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")

        def step(i):
            d = pool[i % len(pool)]
            retrieval.contrastive_step(model, opt, d["eeg"], 1, d["img"], d["txt"], d["labels"], classes, loss_acc, correct)
        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"workload": "configs[1] step, B = 256, exact fp32 products (no split-bf16, no fused transformer block)", "steps": steps,
                "ms_per_step": round(1e3 * dt / steps, 4), "samples_per_s": round(B * steps / dt, 1)}
    finally:
        if old is None:
            os.environ.pop("EEGCLIP_GEMM_PRECISION", None)
        else:
            os.environ["EEGCLIP_GEMM_PRECISION"] = old


def _sec_joint(B=256, steps=24):
    """the joint-subject model (one value embedding per subject, Retrieval/ATMS_retrieval_joint_train.py:172-192) -- the same step on
    batches that MIX the ten subjects (the general case; the reference's own loop feeds one subject per batch, which is the `uniform` line)"""
    from eeg_image_decode_amd import optim, retrieval
    from eeg_image_decode_amd.retrieval_joint import ATMS as JointATMS
    _, _, pool, classes = build(1, 0, B)
    torch.manual_seed(0)
    model = JointATMS(joint_train=True).cuda().train()
    opt = optim.AdamW(model.parameters(), lr=3e-4)
    loss_acc, correct = torch.zeros((), device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(3)
    out = {"workload": "configs[1] step with the joint-subject model (10 value embeddings), B = 256, 1 GPU"}
    for label, ids in (("mixed", [rng.integers(0, 10, B).tolist() for _ in pool]), ("uniform", [int(i) for i in range(len(pool))])):
        def step(i):
            d = pool[i % len(pool)]
            retrieval.contrastive_step(model, opt, d["eeg"], ids[i % len(pool)], d["img"], d["txt"], d["labels"], classes, loss_acc, correct)
        for i in range(24):                                  # (every subject's gradient / optimizer caches exist before the clock starts: one-time
            step(i)                                          #  set-up events of 50-80 ms were seen up to the 18th step of a new id pattern)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[label] = {"steps": steps, "ms_per_step": round(1e3 * dt / steps, 4), "samples_per_s": round(B * steps / dt, 1)}
    eng = model._engine()
    out["fused_block"] = any(k[0] == "f" and "eegclip_token_block_fwd" in pl.op_names() for k, pl in eng.plans.items())
    return out


def _sec_prior_train(B=1024, batches=16):
    from eeg_image_decode_amd.prior import DiffusionPriorUNet, Pipe
    g = torch.Generator().manual_seed(0)
    n = B * batches
    c, h = torch.randn(n, 1024, generator=g).cuda(), torch.randn(n, 1024, generator=g).cuda()
    pipe = Pipe(DiffusionPriorUNet(cond_dim=1024, dropout=0.1), device="cuda")
    dl = [{"c_embedding": c[i:i + B], "h_embedding": h[i:i + B]} for i in range(0, n, B)]       # batches resident in HBM, like the headline
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        pipe.train(dl[:2], num_epochs=1, learning_rate=1e-3)                                    # builds the plans: conditioned ...
        pipe.cond_drop_prob = 1.0
        pipe.train(dl[:2], num_epochs=1, learning_rate=1e-3)                                    # ... and the 10 % of steps that drop the condition
        pipe.cond_drop_prob = 0.1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.train(dl, num_epochs=3, learning_rate=1e-3)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # (16 batches per epoch = the 16,540 training embeddings of THINGS-EEG at batch 1024; the loop's one host sync is the loss readout that ends an epoch)
    return {"workload": "configs[3]: diffusion-prior training step (add_noise, forward, MSE, backward, grad-norm clip, Adam) at batch 1024, 16 batches per epoch, 1 GPU",
            "steps": 3 * batches, "ms_per_step": round(1e3 * dt / (3 * batches), 3), "samples_per_s": round(3 * n / dt, 1)}


def _sec_prior_chain(n=8, steps=50):
    from eeg_image_decode_amd.prior import DiffusionPriorUNet, Pipe
    pipe = Pipe(DiffusionPriorUNet(cond_dim=1024), device="cuda")
    c = torch.randn(n, 1024, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(1)
    pipe.generate(c_embeds=c, num_inference_steps=steps, guidance_scale=5.0, generator=gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        pipe.generate(c_embeds=c, num_inference_steps=steps, guidance_scale=5.0, generator=gen)
    torch.cuda.synchronize()
    return {"workload": f"Pipe.generate: {steps} DDPM steps with classifier-free guidance for {n} EEG embeddings (one captured HIP graph)",
            "ms_per_chain": round(1e3 * (time.perf_counter() - t0) / reps, 2)}


def _sec_cross_attn(images=8):
    from eeg_image_decode_amd.sdxl import cross_attention
    rows = []
    for HW, heads in ((4096, 10), (1024, 20)):
        Bc, C = 2 * images, heads * 64
        q = torch.randn(Bc, HW, C, device="cuda", dtype=torch.float16)
        k, v = torch.randn(Bc, 77, C, device="cuda", dtype=torch.float16), torch.randn(Bc, 77, C, device="cuda", dtype=torch.float16)
        ki, vi = torch.randn(Bc, 4, C, device="cuda", dtype=torch.float16), torch.randn(Bc, 4, C, device="cuda", dtype=torch.float16)
        ms = _ev_ms(lambda: cross_attention(q, k, v, heads, ki, vi, 1.0), 50)
        byts = 2 * q.numel() * 2 + 2 * (k.numel() + ki.numel()) * 2
        rows.append({"HW": HW, "C": C, "us": round(ms * 1e3, 1), "GBs": round(byts / ms / 1e6, 1), "frac_of_hbm_peak": round(byts / ms / 1e6 / PEAK_HBM_GBS, 3)})
    return {"workload": f"configs[4]: softmax(QK^T/8)V + IP-Adapter branch, {images} images x CFG pair, 77 + 4 tokens, fp16; bytes = Q in + O out + K/V",
            "shapes": rows}


def _sec_sdxl_loop():
    from eeg_image_decode_amd import sdxl
    if not hasattr(sdxl, "bench_sampling_loop"):
        return {"skipped": "sampling loop not built in this revision"}
    return sdxl.bench_sampling_loop(images=8, steps=50)


def _sec_end_to_end():
    """the reference's whole input path: THINGS-EEG files on disk -> EEGDataset (float64 -> HBM-resident float32 split) -> shuffled batches of
    256 -> train_model (per-epoch host syncs included).  A synthetic tree in the reference's on-disk format with the real channel / time extents."""
    import shutil
    import tempfile
    from eeg_image_decode_amd import optim, retrieval, synthetic as syn
    from eeg_image_decode_amd.atms import ATMS
    from eeg_image_decode_amd.datasets import EEGDataset
    root = tempfile.mkdtemp(prefix="things_bench_")
    try:
        cfg = syn.write_things_eeg_tree(root, 3, subjects=("sub-01",), channels=63, n_times=300, dt=0.004, train_classes=52, test_classes=200, test_reps=4)
        import contextlib, io
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):           # (the dataset class prints its shapes like the reference; stdout carries ONE json line)
            ds = EEGDataset(cfg["data_path"], subjects=["sub-01"], train=True, config=cfg, features_dir=root)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        torch.manual_seed(0)
        model = ATMS().cuda()
        opt = optim.AdamW(model.parameters(), lr=3e-4)
        ld = ds.loader(batch_size=256, shuffle=True, drop_last=True)
        run = lambda: retrieval.train_model("sub-01", model, ld, opt, "cuda", ds.text_features, ds.img_features, None)
        run()                                                     # first epoch builds the plans
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        epochs = 6
        for _ in range(epochs):
            run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = epochs * len(ld) * 256
        return {"workload": "files -> EEGDataset -> DeviceLoader(256, shuffle) -> train_model, B = 256", "samples": len(ds), "dataset_build_s": round(t_build, 3),
                "epochs": epochs, "batches_per_epoch": len(ld), "samples_per_s": round(n / dt, 1), "ms_per_step": round(1e3 * dt / (epochs * len(ld)), 4)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU (BASELINE configs[1]: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (other SURVEY section 8 rows)")
    ap.add_argument("--breakdown", action="store_true", help="time every kernel of the fwd/bwd plans with HIP events and print a table")
    ap.add_argument("--roofline-kernel", default="auto", help="kernel family for the roofline object: gemm_bf16x3, gemm_f32 or an op name such as "
                    "eegclip_sconv_fwd (auto = the family with the largest total time)")
    args = ap.parse_args()
    if args.breakdown:
        os.environ["EEGCLIP_STEP_PLAN"] = "0"       # every op is timed through the encoder's own plans (the step plan replays the same launches)

    from eeg_image_decode_amd import dist as edist
    from eeg_image_decode_amd import retrieval
    rank, local_rank, world = edist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    B = args.batch
    model, opt, pool, classes = build(world, rank, B)
    loss_acc = []           # per-step loss scalars (the product loop's form, retrieval.train_model): summed after the timed region
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
    # (several ranks: every step holds collectives, so all ranks must run the SAME number of steps -- a fixed count, not a time-based one)
    while (time.perf_counter() - t_ramp < 0.4 or i_ramp < 3) if world == 1 else i_ramp < 48:
        step(i_ramp)
        i_ramp += 1
        if i_ramp % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    # what the product's own loops do before their first step (retrieval.train_model, Pipe.train): one full collection + gc.freeze(), so that no
    # generation-2 collection -- 75 ms on this host, at an unpredictable step (tools/host_stalls.py) -- lands in an epoch.  Before the warm-up, not in the clock.
    retrieval.settle_gc()
    for i in range(args.warmup):
        step(i)
    eng = model._engine()
    plans = {k: v for k, v in eng.plans.items()}
    # ---- the dominant kernel: 3 instrumented single-stream steps (outside the headline clock) time every launch on its own; launches are
    # grouped into families (all GEMM launches of one arithmetic are one kernel template) and the family with the largest total time is
    # the one the roofline object describes.  ALL its launches are then timed live inside the timed region (HIP events on the stream each
    # launch goes to -- the backward overlaps two streams there, which stretches per-launch durations: both figures are reported).
    # (the single-submission step plan -- step_plan.py -- replays the very same launches; the instrumented steps go launch by launch through the
    #  encoder's own plans, whose ops carry the timing hooks)
    os.environ["EEGCLIP_STEP_PLAN_OFF_FOR_BENCH"] = os.environ.get("EEGCLIP_STEP_PLAN", "1")
    os.environ["EEGCLIP_STEP_PLAN"] = "0"
    for k, pl in plans.items():
        pl.use_side_stream = False
        pl.time_ops(range(len(pl.ops)))
    for i in range(3):
        step(i)
    single = {}                                   # (plan key, op index) -> mean ms on one stream
    for k, pl in plans.items():
        for idx, v in pl.timings_ms().items():
            single[(k, idx)] = float(np.mean(v))
        pl.time_ops([])
        pl.use_side_stream = True
    os.environ["EEGCLIP_STEP_PLAN"] = os.environ.pop("EEGCLIP_STEP_PLAN_OFF_FOR_BENCH")
    step(0)                                       # (back on the step plan, if the configuration has one)
    sp = next(iter(retrieval.step_plans_of(model)), None)
    fam_ops, fam_ms = {}, {}
    for (k, idx), ms in single.items():
        name = plans[k].ops[idx][2]
        d = _desc_of(plans[k], idx)
        if algorithmic_cost(name, d, B) is None:
            continue
        fam = family_of(name, d)
        fam_ops.setdefault(fam, []).append((k, idx))
        fam_ms[fam] = fam_ms.get(fam, 0.0) + ms
    dominant = max(fam_ms, key=fam_ms.get) if args.roofline_kernel == "auto" else args.roofline_kernel
    if args.breakdown:
        for k, pl in plans.items():
            pl.use_side_stream = False
            pl.time_ops(range(len(pl.ops)))
    else:
        by_plan = {}
        for k, idx in fam_ops.get(dominant, []):
            by_plan.setdefault(k, []).append(idx)
        # every launch of the family is timed on every TIME_EVERY-th step of the timed region (two HIP events per launch: ~40 per step for the GEMM
        # family, ~0.1 ms of a 1.2 ms step if recorded on every step -- instrumentation the product does not carry; 3+ sampled steps x 20 launches)
        every = TIME_EVERY if args.steps >= 2 * TIME_EVERY else 1
        if sp is not None:                        # the step plan holds the encoder plans' ops at fixed offsets
            sp.pl.time_ops([sp.index_of(k[0], i) for k, idxs in by_plan.items() for i in idxs], every=every)
        else:
            for k, idxs in by_plan.items():
                plans[k].time_ops(idxs, every=every)

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
    final_loss = float(retrieval.running_loss(loss_acc)) / max(1, len(loss_acc))

    roof = None
    if args.breakdown:
        rows = []
        for k, pl in plans.items():
            for idx, v in pl.timings_ms().items():
                name = pl.ops[idx][2]
                d = _desc_of(pl, idx)
                tag = (name if d is None else f"{name}[part {d}]" if isinstance(d, int) else f"{name}[{d.label}/s{d.slices}]" if hasattr(d, "label")
                       else f"{name}[{d.M}x{d.N}x{d.K}/s{d.slices}]" if hasattr(d, "slices")
                       else f"{name}[{d.M}x{d.N}x{d.K}{'/sk' + str(d.split_k) if d.split_k > 1 else ''}]")
                rows.append((float(np.mean(v[-args.steps:])), k[0], idx, tag))
        rows.sort(reverse=True)
        if rank == 0:
            tot = sum(r[0] for r in rows)
            print(f"# per-kernel HIP-event breakdown (ms/step, mean over {args.steps} steps); sum of kernels = {tot:.3f} ms, step = {ms_per_step:.3f} ms", file=sys.stderr)
            for ms, ph, idx, tag in rows:
                print(f"#  {ms:8.4f} ms  {100 * ms / tot:5.1f}%  {ph}{idx:02d}  {tag}", file=sys.stderr)
    elif dominant in fam_ops:
        ops = fam_ops[dominant]
        live = {}
        if sp is not None:
            tm = sp.pl.timings_ms()
            for (k, idx) in ops:
                live[(k, idx)] = float(np.mean(tm[sp.index_of(k[0], idx)][-args.steps:]))
        else:
            for k in {k for k, _ in ops}:
                for idx, v in plans[k].timings_ms().items():
                    live[(k, idx)] = float(np.mean(v[-args.steps:]))
        work = {o: algorithmic_cost(plans[o[0]].ops[o[1]][2], _desc_of(plans[o[0]], o[1]), B) for o in ops}
        bound, unit = work[ops[0]][0], work[ops[0]][2]
        w_tot = sum(w[1] for w in work.values())
        ms_live, ms_single = sum(live[o] for o in ops), sum(single[o] for o in ops)
        peak, scale, u = family_peak(dominant, bound)
        ach = w_tot / (ms_live * scale)
        big = max(ops, key=lambda o: work[o][1])
        dbig = _desc_of(plans[big[0]], big[1])
        traffic, tsrc = pmc_traffic(dominant, B)
        kernel_names = {"conv_stack": "eeg::cstack_{stats1,fwd,bwd<false>,bwd<true>,bwd_w2}_kernel (+ weight packs, row reductions): Conv(1x25) + AvgPool + BatchNorm + ELU + "
                                      "Conv(63x1) forward and backward recomputed from the token rows on the bf16 matrix cores (split-bf16 products); y1 / dy1 never in HBM",
                        "gemm_bf16x3": "eeg::head_gemm_kernel + eeg::gemm_x3_kernel + eeg::wgrad_tok_kernel (the Linears outside the fused transformer block: head forward / "
                                       "dX / query gradient K-parallel from planes on head_gemm, the head's weight gradients on gemm_x3, the block's weight gradients "
                                       "from token planes on wgrad_tok + its slab reduction; split-bf16 products)",
                        "gemm_f32": "eeg::gemm_f32_fast_kernel (every Linear of the step, exact fp32 products)",
                        "token_block": "eeg::token_block_{fwd,bwd_a,bwd_b}_kernel (the encoder's transformer block, one workgroup per sample: forward and the "
                                       "two backward parts; bytes = activations that must cross HBM for the batch-wide weight-gradient GEMMs)"}
        roof = {"kernel": kernel_names.get(dominant, dominant), "launches_per_step": len(ops), "bound": bound, "achieved": round(ach, 2), "peak": round(peak, 1),
                "unit": u, "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc,
                "avg_launch_ms": round(ms_live / len(ops), 5),
                "timed_steps": f"every {TIME_EVERY if args.steps >= 2 * TIME_EVERY else 1}. step of the timed region, all launches of the family",
                "timing": "HIP events stamped with each kernel's own begin / end timestamps (hipExtLaunchKernel start / stop events on the launch stream; "
                          "EEGCLIP_TIMING=bracket: event records around the launch, ~5 us more per launch)", "algorithmic_work_per_launch": w_tot / len(ops), "work_unit": unit,
                "share_of_kernel_time_single_stream": round(fam_ms[dominant] / sum(single.values()), 3),
                "single_stream": {"achieved": round(w_tot / (ms_single * scale), 2), "frac": round(w_tot / (ms_single * scale) / peak, 4),
                                  "note": "same launches timed one at a time on one stream (3 instrumented steps before the timed region)"},
                "largest_launch": {"shape": f"{dbig.M}x{dbig.N}x{dbig.K}" if hasattr(dbig, "M") else getattr(dbig, "label", plans[big[0]].ops[big[1]][2]), "avg_ms": round(live[big], 5),
                                   "achieved": round(work[big][1] / (live[big] * scale), 2), "frac": round(work[big][1] / (live[big] * scale) / peak, 4)}}
        if bound == "mfma":
            roof["frac_of_f32_mfma_peak"] = round(ach / PEAK_F32_MFMA_TF, 4)
        if hasattr(dbig, "M") and hasattr(dbig, "split_k"):
            # what the HIP-event bracketing itself adds to a launch (marker packets + dispatch gaps on both sides): the largest launch 40 times
            # between ONE event pair against 40 individually bracketed launches, live, on an idle GPU after the timed region.  rocprofv3's kernel
            # timestamps (profiles/r2_final_kernel_stats.csv, same command) agree with the NET figure, not with the raw event time.
            import ctypes
            L_, st_ = plans[big[0]].L, torch.cuda.current_stream().cuda_stream
            ref_ = ctypes.byref(dbig)
            for _ in range(5):
                L_.eegclip_gemm_f32(ref_, st_)
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(40):
                L_.eegclip_gemm_f32(ref_, st_)
            b_.record()
            pairs = []
            for _ in range(40):
                x_, y_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                x_.record()
                L_.eegclip_gemm_f32(ref_, st_)
                y_.record()
                pairs.append((x_, y_))
            torch.cuda.synchronize()
            t_batch = a_.elapsed_time(b_) / 40.0
            t_br = float(np.mean([x_.elapsed_time(y_) for x_, y_ in pairs]))
            ov = max(0.0, t_br - t_batch)
            net_single = max(1e-9, ms_single - ov * len(ops))
            roof["event_bracket_overhead_ms_per_launch"] = round(ov, 5)       # (what a record-around-the-launch bracket would add; informational)
            if not plans[big[0]].kernel_timestamps:                             # EEGCLIP_TIMING=bracket: the figures above carry that overhead
                roof["single_stream"]["achieved_net_of_event_overhead"] = round(w_tot / (net_single * scale), 2)
                roof["single_stream"]["frac_net_of_event_overhead"] = round(w_tot / (net_single * scale) / peak, 4)
            roof["largest_launch"]["back_to_back_ms"] = round(t_batch, 5)
            roof["largest_launch"]["back_to_back_achieved"] = round(work[big][1] / (t_batch * scale), 2)
            roof["largest_launch"]["back_to_back_frac"] = round(work[big][1] / (t_batch * scale) / peak, 4)
        if dominant == "gemm_bf16x3":
            roof["peak_note"] = "2.5 PFLOP/s dense bf16 MFMA / 3 products per multiply-add; achieved counts algorithmic 2MNK flops"
        # every priced family against ITS roofline (single-stream launch times of the 3 instrumented steps: one launch at a time, nothing co-running)
        fams = {}
        for f, fops in fam_ops.items():
            fw = [algorithmic_cost(plans[o[0]].ops[o[1]][2], _desc_of(plans[o[0]], o[1]), B) for o in fops]
            fpeak, fscale, fu = family_peak(f, fw[0][0])
            fa = sum(x[1] for x in fw) / (fam_ms[f] * fscale)
            fams[f] = {"bound": fw[0][0], "launches_per_step": len(fops), "ms_per_step_single_stream": round(fam_ms[f], 4), "achieved": round(fa, 2), "peak": round(fpeak, 1),
                       "unit": fu, "frac": round(fa / fpeak, 4)}
            # algorithmic bytes per launch (family mean) next to the PMC traffic of the committed profile, their ratio, and the matrix-pipe busy fraction:
            # traffic well above the algorithmic bytes = wasted re-reads; a GEMM family is also priced against the HBM roof (the K ~ 16 k weight
            # gradients are as much HBM- as MFMA-limited): frac_of_binding_roof = time the binding roof allows / time taken
            ab = [algorithmic_bytes(plans[o[0]].ops[o[1]][2], _desc_of(plans[o[0]], o[1]), B) for o in fops]
            if all(x is not None for x in ab):
                fams[f]["algorithmic_bytes_per_launch"] = round(sum(ab) / len(fops))
                tr, _ = pmc_traffic(f, B)
                if tr:
                    fams[f]["traffic_per_launch"] = tr
                    fams[f]["traffic_over_algorithmic"] = round(tr / max(1.0, sum(ab) / len(fops)), 2)
                if fw[0][0] == "mfma":
                    t_mfma = sum(x[1] for x in fw) / (fpeak * 1e12) * 1e3          # ms at the MFMA roof
                    t_hbm = sum(ab) / (PEAK_HBM_GBS * 1e9) * 1e3                   # ms at the HBM roof
                    fams[f]["frac_of_hbm_roof"] = round(t_hbm / fam_ms[f], 4)
                    fams[f]["frac_of_binding_roof"] = round(max(t_mfma, t_hbm) / fam_ms[f], 4)
            if fw[0][0] == "hbm":
                fl = [algorithmic_flops(plans[o[0]].ops[o[1]][2], _desc_of(plans[o[0]], o[1]), B) for o in fops]
                if all(x is not None for x in fl) and sum(fl) > 0:                 # the same launches against the (split-bf16) matrix-pipe roof
                    fams[f]["frac_of_mfma_roof"] = round(sum(fl) / (fam_ms[f] * 1e-3) / (PEAK_BF16_MFMA_TF / 3.0 * 1e12), 4)
            mb = pmc_mfma_busy(f, B)
            if mb is not None:
                fams[f]["mfma_busy_frac"] = mb
        roof["families"] = dict(sorted(fams.items(), key=lambda kv: -kv[1]["ms_per_step_single_stream"]))
        # the headline family is simply the one with the most single-stream kernel time; the three largest with BOTH roofs side by side, so that a change
        # of ruler from round to round is visible at the top level (no tie rule)
        for k_ in ("frac_of_mfma_roof", "frac_of_hbm_roof", "frac_of_binding_roof"):
            if k_ in fams[dominant]:
                roof[k_] = fams[dominant][k_]
        roof["largest_families"] = [{"family": f, **{k_: v[k_] for k_ in ("ms_per_step_single_stream", "bound", "frac", "frac_of_mfma_roof", "frac_of_hbm_roof") if k_ in v}}
                                    for f, v in list(roof["families"].items())[:3]]

    distributed = None
    if world > 1:
        # what the run actually was (the driver's first multi-GPU line must show "RCCL saw N ranks") + where the step's communication time goes:
        # 5 extra instrumented steps AFTER the timed region, an event pair on the compute stream around every collective (async ones: issue .. wait())
        distributed = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                       "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if torch.distributed.get_backend() == "nccl" else None,
                       "devices_visible": torch.cuda.device_count(), "sync_batchnorm": bool(model.sync_batchnorm),
                       "loss_mode": {"local_loss": model.loss_func.local_loss, "gather_with_grad": model.loss_func.gather_with_grad}}
        log = _CollectiveLog()
        with log:
            for i in range(5):
                step(i)
            torch.cuda.synchronize()
        distributed["collectives"] = log.summary(5)
        barrier()

    out = {
        "metric": "EEG-CLIP contrastive train samples/sec (global batch)", "value": round(value, 1), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",        # fp32 tensors end to end; see config.gemm_arithmetic
        "config": {"workload": "configs[1]: full ATM-S EEG encoder (63ch x 250t -> 1024-d) contrastive train step vs frozen 1024-d CLIP "
                               "embeddings, 256 samples/GPU" + (f", global batch {world * B} with RCCL all-gather negatives (configs[2])" if world > 1 else ""),
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}", "n_classes_for_accuracy": N_CLASSES,
                   "optimizer": "AdamW lr 3e-4 (fused)", "final_mean_loss": round(final_loss, 4),
                   "gemm_arithmetic": "bf16x3 split products, fp32 accumulate (embeddings within 3e-5 of exact fp32 products)"
                   if os.environ.get("EEGCLIP_GEMM_PRECISION", "bf16x3") != "f32" else "exact fp32 products (v_mfma_f32_16x16x4_f32)",
                   "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 4),
                   "submission": ("one eegclip_plan_run call per step (step_plan.py: forward, accuracy readout, fused InfoNCE, backward, AdamW)" if sp is not None
                                  else "launch by launch from the Python loop (two encoder plans + loss / optimizer calls)"),
                   "launches_per_step": (len([o for o in sp.pl.ops if o[0] is not None]) if sp is not None else None),
                   "python_gc": "gc.collect() + gc.freeze() once before the warm-up, as the product's training loops do (retrieval.settle_gc): a "
                                "generation-2 collection costs 75 ms on this host"},
        "roofline": roof,
    }
    if distributed is not None:
        out["distributed"] = distributed
    if rank == 0 and world == 1 and not args.no_secondary:
        del model, opt, pool
        torch.cuda.empty_cache()
        out["secondary"] = secondary()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


class _WorkProxy:
    """an async collective's Work handle whose wait() also stamps the end event (pybind Work objects take no new attributes)"""

    def __init__(self, work, end_event):
        self._work, self._end = work, end_event

    def wait(self, *a, **k):
        r = self._work.wait(*a, **k)
        self._end.record()
        return r

    def __getattr__(self, name):
        return getattr(self._work, name)


class _CollectiveLog:
    """wraps torch.distributed's collectives while active: per call the kind, payload bytes and an event pair on the current (compute) stream --
    around the call for blocking collectives, from issue to .wait() for async ones -- i.e. the time the compute stream is held, not wire time"""
    KINDS = ("all_reduce", "all_gather_into_tensor", "reduce_scatter_tensor")

    def __init__(self, event_factory=None):
        self.new_event = event_factory or (lambda: torch.cuda.Event(enable_timing=True))

    def __enter__(self):
        import torch.distributed as dist
        self.dist, self.saved, self.calls = dist, {}, []
        for kind in self.KINDS:
            real = getattr(dist, kind)
            self.saved[kind] = real
            setattr(dist, kind, self._wrap(kind, real))
        return self

    def _wrap(self, kind, real):
        log = self

        def call(*a, **k):
            t = a[0]
            e0, e1 = log.new_event(), log.new_event()
            e0.record()
            work = real(*a, **k)
            is_async = bool(k.get("async_op", False))
            log.calls.append({"kind": kind, "bytes": t.numel() * t.element_size(), "async": is_async, "ev": (e0, e1)})
            if work is not None and is_async:
                return _WorkProxy(work, e1)
            e1.record()
            return work
        return call

    def __exit__(self, *exc):
        for kind, real in self.saved.items():
            setattr(self.dist, kind, real)

    def summary(self, steps):
        by = {}
        for c in self.calls:
            key = f"{c['kind']}{'(async)' if c['async'] else ''}:{c['bytes']}B"
            try:
                us = 1e3 * c["ev"][0].elapsed_time(c["ev"][1])
            except Exception:                                     # an async collective whose wait() never ran
                us = None
            by.setdefault(key, []).append(us)
        rows = {k: {"per_step": round(len(v) / steps, 2), "mean_us_stream_held": (round(float(np.mean([x for x in v if x is not None])), 1) if any(x is not None for x in v) else None)}
                for k, v in by.items()}
        # payload and algorithmic bandwidth (payload bytes / time the compute stream was held; all_gather / reduce_scatter: the FULL gathered / scattered
        # tensor) per collective, so that a multi-GPU line can be read against DESIGN.md section 7's per-collective predictions without a profiler
        for k, r in rows.items():
            nbytes = int(k.rsplit(":", 1)[1][:-1])
            r["bytes"] = nbytes
            r["algbw_GBps"] = round(nbytes / (r["mean_us_stream_held"] * 1e3), 2) if r["mean_us_stream_held"] else None
        held = sum(r["per_step"] * r["mean_us_stream_held"] for k, r in rows.items() if r["mean_us_stream_held"] is not None and "(async)" not in k)
        return {"per_step": round(len(self.calls) / steps, 2), "by_kind_and_payload": rows, "sum_us_stream_held_per_step": round(held, 1),
                "note": "blocking collectives: event pair around the call on the compute stream; async ones: issue .. wait() (mostly overlapped work, not exposure)"}


def _desc_of(plan, idx):
    fn, a, name = plan.ops[idx][:3]
    if name in ("eegclip_gemm_f32", "eegclip_head_gemm"):
        return a[0]._obj
    if name == "eegclip_token_block_bwd":
        return int(a[1])                                  # which part of the fused backward
    if name.startswith("eegclip_wgrad_tok"):
        import types
        probs, n, B = a[0], int(a[1]), int(a[2])
        return types.SimpleNamespace(flops=sum(2.0 * probs[i].M * probs[i].N * 64 * B for i in range(n)), label="+".join(f"{probs[i].M}x{probs[i].N}x{64 * B}" for i in range(n)),
                                     slices=int(a[3]),
                                     bytes=sum((probs[i].m_groups + 1) * B * 65536 + 4.0 * probs[i].M * probs[i].N for i in range(n)))
    return None


if __name__ == "__main__":
    main()
