"""configs[2] on the code every rank runs: the FUSED row-sharded InfoNCE (eeg_image_decode_amd/loss.py `sharded_blocks`, csrc/infonce_fused.hip) at
sizes where `fused_enabled` is true -- n per rank a multiple of 64, D = 1024 -- against

  * tests/golden/dist_loss_fused.npz: per-rank loss, d loss / d features and d loss / d logit_scale recorded from THE REFERENCE's ClipLoss
    (models/loss.py:20-141) under gloo with (world, n) = (2, 64), (4, 64) and (8, 256) [= configs[2]: 8 x 256 rows, N = 2048], all three gather modes;
  * an fp64 evaluation of models/loss.py:113-115,129-140 for the row-sharded blocks themselves (loss, both gradient matrices, d scale).

Two drivers: every rank of the job simulated one after the other on ONE device through `sharded_blocks` (the collectives around it are plain sums
here) -- runs on the CPU lane emulator at (2, 64) and on the MI355X at every size; and real processes over gloo sharing the GPU (the all-gather /
reduce-scatter calls themselves)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, SEED
from eeg_image_decode_amd import synthetic as syn

S0 = float(np.log(1 / 0.07))
MODES = [(False, False), (False, True), (True, True)]


def _features(world, n):
    a_all = syn.unit_features(SEED + 11, n * world, tag="da") * 32.0            # must match tests/golden/make_golden.py:_dist_worker_big
    b_all = syn.unit_features(SEED + 11, n * world, tag="db")
    return a_all.astype(np.float32), b_all.astype(np.float32)


def _simulate_job(ploss, dev, world, n, mode, planes=2):
    """every rank's `sharded_blocks` in turn on one device; all-gather = the full matrices, reduce-scatter(sum) = a sum over the simulated ranks"""
    a_np, b_np = _features(world, n)
    a_all, b_all = torch.from_numpy(a_np).to(dev), torch.from_numpy(b_np).to(dev)
    sc = torch.full((1,), S0, dtype=torch.float32, device=dev)
    per_rank, ga_sum, gb_sum = [], None, None
    calls = []
    real = ploss.fused_infonce
    ploss.fused_infonce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        for r in range(world):
            acc = torch.zeros(2, dtype=torch.float32, device=dev)
            sl = slice(r * n, (r + 1) * n)
            da, ga, dbs, gbs = ploss.sharded_blocks(mode[0], mode[1], r, world, a_all[sl].contiguous(), [b_all[sl].contiguous()], a_all, [b_all], (1.0,),
                                                    sc, acc, True, [True], True, planes)
            per_rank.append((acc.clone(), da, dbs[0]))
            if ga is not None:
                ga_sum = ga.clone() if ga_sum is None else ga_sum + ga
            if gbs[0] is not None:
                gb_sum = gbs[0].clone() if gb_sum is None else gb_sum + gbs[0]
    finally:
        ploss.fused_infonce = real
    assert len(calls) == world, "the fused route was not taken"
    out = []
    for r, (acc, da, db) in enumerate(per_rank):
        sl = slice(r * n, (r + 1) * n)
        if ga_sum is not None:
            da = da + ga_sum[sl]
        if gb_sum is not None:
            db = db + gb_sum[sl]
        out.append((float(acc[0]), da.cpu().numpy(), db.cpu().numpy(), float(acc[1])))
    return out


def _simulate_training_job(ploss, dev, world, n, planes=2):
    """the TRAINING configuration (local_loss + gather_with_grad, frozen targets): sharded_blocks takes the all-planes route (round 6); the gradient of a
    rank's queries = its own part + its rows of the reduce-scattered gathered-copy gradients"""
    a_np, b_np = _features(world, n)
    a_all, b_all = torch.from_numpy(a_np).to(dev), torch.from_numpy(b_np).to(dev)
    sc = torch.full((1,), S0, dtype=torch.float32, device=dev)
    per_rank, ga_sum = [], None
    calls = []
    real = ploss._sharded_blocks_on_planes
    ploss._sharded_blocks_on_planes = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        for r in range(world):
            acc = torch.zeros(2, dtype=torch.float32, device=dev)
            sl = slice(r * n, (r + 1) * n)
            da, ga, dbs, gbs = ploss.sharded_blocks(True, True, r, world, a_all[sl].contiguous(), [b_all[sl].contiguous()], a_all, [b_all], (1.0,), sc, acc, True,
                                                    [False], True, planes)
            assert dbs == [None] and gbs == [None]
            per_rank.append((acc.clone(), da))
            ga_sum = ga.clone() if ga_sum is None else ga_sum + ga
    finally:
        ploss._sharded_blocks_on_planes = real
    assert len(calls) == world, "the all-planes route was not taken"
    return [(float(acc[0]), (da + ga_sum[r * n:(r + 1) * n]).cpu().numpy(), None, float(acc[1])) for r, (acc, da) in enumerate(per_rank)]


def _check_against_the_reference_fixture(out, world, n, mode):
    g = np.load(os.path.join(GOLDEN, "dist_loss_fused.npz"))
    tag = f"w{world}_n{n}_ll{int(mode[0])}_gwg{int(mode[1])}"
    cols = g[tag + "_da"].shape[2]
    for r in range(world):
        loss, da, db, ds = out[r]
        assert abs(loss - g[tag + "_loss"][r]) < 1e-4, (r, loss, g[tag + "_loss"][r])
        ra, rb = g[tag + "_da"][r], g[tag + "_db"][r]
        np.testing.assert_allclose(da[:, :cols], ra, atol=5e-4 * np.abs(ra).max(), err_msg=f"{tag} rank {r} da")
        if db is not None:
            np.testing.assert_allclose(db[:, :cols], rb, atol=5e-4 * np.abs(rb).max(), err_msg=f"{tag} rank {r} db")
        assert abs(ds - g[tag + "_ds"][r]) < 2e-4 * abs(g[tag + "_ds"][r]), (r, ds, g[tag + "_ds"][r])


# ------------------------------------------------------------------------------------------------ host logic + kernels on the CPU lane emulator
@pytest.mark.emu
@pytest.mark.parametrize("mode", MODES)
def test_simulated_two_rank_job_on_the_emulator_matches_the_reference_gloo_fixture(mode):
    from emu_patch import product_on_emulator
    with product_on_emulator():
        from eeg_image_decode_amd import loss as ploss
        out = _simulate_job(ploss, "cpu", 2, 64, mode)
    _check_against_the_reference_fixture(out, 2, 64, mode)


@pytest.mark.emu
def test_training_configuration_on_planes_on_the_emulator_matches_the_reference_gloo_fixture():
    from emu_patch import product_on_emulator
    with product_on_emulator():
        from eeg_image_decode_amd import loss as ploss
        out = _simulate_training_job(ploss, "cpu", 2, 64)
    _check_against_the_reference_fixture(out, 2, 64, (True, True))


# ------------------------------------------------------------------------------------------------ MI355X
@pytest.mark.gpu
@pytest.mark.parametrize("planes", [2, 1])
@pytest.mark.parametrize("world,n", [(2, 64), (4, 64), (8, 256)])
def test_training_configuration_on_planes_on_the_gpu_matches_the_reference_gloo_fixture(world, n, planes):
    """frozen targets, local_loss + gather_with_grad: gradient matrices as planes, the swapped blocks' produced transposed, both gradient GEMMs K-parallel from
    planes (round 6) -- every rank of the job against the reference's own ClipLoss under gloo"""
    from eeg_image_decode_amd import loss as ploss
    out = _simulate_training_job(ploss, "cuda", world, n, planes)
    if planes == 2:
        _check_against_the_reference_fixture(out, world, n, (True, True))
    else:       # one bf16 product per multiply-add: the throughput arithmetic, within its documented distance of the reference
        g = np.load(os.path.join(GOLDEN, "dist_loss_fused.npz"))
        tag = f"w{world}_n{n}_ll1_gwg1"
        for r in range(world):
            assert abs(out[r][0] - g[tag + "_loss"][r]) < 5e-3
            ra = g[tag + "_da"][r]
            np.testing.assert_allclose(out[r][1][:, :ra.shape[1]], ra, atol=3e-2 * np.abs(ra).max())



@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("world,n", [(2, 64), (4, 64), (8, 256)])
def test_simulated_job_on_the_gpu_matches_the_reference_gloo_fixture(world, n, mode):
    """(8, 256) is configs[2] itself: every one of the 8 ranks' blocks (col0 = 256 r) with the LDS-DMA / counted-vmcnt kernels the GPU runs"""
    from eeg_image_decode_amd import loss as ploss
    out = _simulate_job(ploss, "cuda", world, n, mode)
    _check_against_the_reference_fixture(out, world, n, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("planes", [2, 1])
@pytest.mark.parametrize("rank", [0, 3, 7])
def test_row_sharded_blocks_at_the_per_rank_shape_against_fp64(rank, planes):
    """n = 256, N = 2048, D = 1024, col0 = 256 r: both blocks (A_r, B_all) / (B_r, A_all) of one launch; loss, G1, G2 and d scale against an fp64
    evaluation of models/loss.py:113-115,129-140 (parity arithmetic against exact products; throughput arithmetic against the products of the
    bf16-rounded features, which is what it defines, and within its documented budget of the exact ones)"""
    from eeg_image_decode_amd import loss as ploss
    from test_kernels_gemm_x3 import bf16_round
    W, n, Dm = 8, 256, 1024
    N = W * n
    a_np, b_np = _features(W, n)
    a_all, b_all = torch.from_numpy(a_np).cuda(), torch.from_numpy(b_np).cuda()
    sl = slice(rank * n, (rank + 1) * n)
    ap_all, bp_all = ploss.split_planes(a_all, planes), ploss.split_planes(b_all, planes)
    cut = lambda p: (p[0][sl], p[1][sl] if planes == 2 else None)
    sc = torch.full((1,), S0, dtype=torch.float32, device="cuda")
    acc = torch.zeros(2, dtype=torch.float32, device="cuda")
    w = 0.5 * 0.99
    G1, G2 = ploss.fused_infonce([(cut(ap_all), bp_all, rank * n, w), (cut(bp_all), ap_all, rank * n, w)], n, N, Dm, planes, n, sc, acc,
                                 [(0, None), (1, None)])
    torch.cuda.synchronize()
    A64, B64 = a_np.astype(np.float64), b_np.astype(np.float64)
    if planes == 1:
        A64, B64 = bf16_round(a_np).astype(np.float64), bf16_round(b_np).astype(np.float64)
    want_loss, want_ds = 0.0, 0.0
    for (Q, K), G in (((A64[sl], B64), G1), ((B64[sl], A64), G2)):
        raw = Q @ K.T
        S = S0 * raw
        m = S.max(1, keepdims=True)
        lse = (m + np.log(np.exp(S - m).sum(1, keepdims=True)))[:, 0]
        pos = S[np.arange(n), rank * n + np.arange(n)]
        want_loss += w / n * (lse - pos).sum()
        P = np.exp(S - lse[:, None])
        P[np.arange(n), rank * n + np.arange(n)] -= 1.0
        Gp = w / n * P
        want_ds += (Gp * raw).sum()
        np.testing.assert_allclose(G.cpu().numpy(), S0 * Gp, atol=3e-4 * np.abs(S0 * Gp).max())
    assert abs(float(acc[0]) - want_loss) < (1e-4 if planes == 2 else 3e-4), (float(acc[0]), want_loss)
    assert abs(float(acc[1]) - want_ds) < 2e-4 * abs(want_ds) + 1e-6, (float(acc[1]), want_ds)
    if planes == 1:
        # the throughput arithmetic's distance to the exact logits: bf16 rounding of both feature matrices (|logit| <= 2.66 * 32)
        exact = S0 * (a_np[sl].astype(np.float64) @ b_np.astype(np.float64).T)
        assert np.abs(S0 * (A64[sl] @ B64.T) - exact).max() < 6e-2


def _loss_worker(rank, world, port, mode, n, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    a_np, b_np = _features(world, n)
    a = torch.from_numpy(a_np[rank * n:(rank + 1) * n]).cuda().requires_grad_(True)
    b = torch.from_numpy(b_np[rank * n:(rank + 1) * n]).cuda().requires_grad_(True)
    sc = torch.tensor(S0, device="cuda", requires_grad=True)
    from eeg_image_decode_amd import loss as ploss
    calls = []
    real = ploss.fused_infonce
    ploss.fused_infonce = lambda *a_, **k: (calls.append(1), real(*a_, **k))[1]
    loss = ploss.ClipLoss(local_loss=mode[0], gather_with_grad=mode[1], rank=rank, world_size=world)(a, b, sc)
    loss.backward()
    torch.cuda.synchronize()
    ret[rank] = (float(loss), a.grad.cpu().numpy(), b.grad.cpu().numpy(), float(sc.grad), len(calls))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("world,n", [(2, 64), (4, 64), (8, 256)])
def test_clip_loss_processes_on_the_fused_route_match_the_reference_gloo_fixture(world, n, mode):
    """B2 with the collectives: `world` processes share the GPU through gloo (RCCL refuses duplicate devices); the fused route is asserted.
    (8, 256): eight ranks x 256 rows, N = 2048 -- configs[2] in everything but the transport."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29901 + 16 * world + 2 * int(mode[0]) + int(mode[1]) + (8 if n == 256 else 0)
    mp.spawn(_loss_worker, args=(world, port, mode, n, ret), nprocs=world, join=True)
    assert all(ret[r][4] == 1 for r in range(world)), "the fused route was not taken"
    _check_against_the_reference_fixture([ret[r][:4] for r in range(world)], world, n, mode)
