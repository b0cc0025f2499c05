"""Parity at the sizes BASELINE.json's configs name (VERDICT r1 "configs not exercised at their size"):
  configs[1]  train-mode forward + backward + AdamW at B = 256 against the oracle (dropout off, and with the Philox masks shared)
  configs[2]  the three ClipLoss gather modes with 2 and 4 ranks on the GPU against the fixture recorded from the reference under gloo
  configs[3]  diffusion-prior forward / backward / Pipe.train step at batch 1024, its 2-rank data-parallel step, 650 steps of the lr schedule
  configs[4]  the SDXL cross-attention kernel at 8 images x CFG pair, 64x64 / 32x32 latents (C = 640 / 1280), fp16 and bf16
Ranks > 1 share the one GPU of the test box through gloo (RCCL refuses duplicate devices); everything but the transport is production code."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, SEED
from eeg_image_decode_amd import synthetic as syn
from oracle import atms as oatms
from oracle import loops as oloops
from oracle import prior as oprior
from oracle import sdxl_attn
from philox_np import keep_mask
from test_model_gpu import T, _check_grads, make_model, zero_dropout

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------------------ configs[1]
def test_train_step_at_batch_256_matches_the_oracle():
    """the bench configuration itself: one full step (forward train mode, image + text InfoNCE, backward -- split-K factors, conv weight-gradient
    tilings and the second stream all depend on B --, fused AdamW) against the oracle's autograd and its AdamW"""
    from eeg_image_decode_amd import optim
    state_np = syn.make_state(SEED, oatms.state_spec())
    B = 256
    x = T(syn.eeg_batch(SEED + 70, B))
    img, txt = T(syn.unit_features(SEED + 70, B, tag="img")), T(syn.unit_features(SEED + 70, B, tag="txt"))
    m = make_model(state_np)
    zero_dropout(m)
    m.train()
    opt = optim.AdamW(m.parameters(), lr=3e-4)
    z = m(x.cuda(), 1)
    loss = m.loss_func.forward_mixed(z, [(img.cuda(), 0.99), (txt.cuda(), 0.01)], m.logit_scale)
    loss.backward()
    tr = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    lo, zo, grads, _ = tr.loss_and_grads(x, torch.full((B,), 1).long(), img, txt, train=True)
    np.testing.assert_allclose(z.detach().cpu().numpy(), zo.numpy(), atol=1e-4)
    assert abs(float(loss) - float(lo)) < 1e-4
    _check_grads(m, {k: (v.numpy() if v is not None else None) for k, v in grads.items()}, atol_rel=3e-3)
    opt.step()
    tr2 = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    tr2.step(x, torch.full((B,), 1).long(), img, txt)
    for k, p in m.named_parameters():
        if p.grad is None or k in oloops.ZERO_GRAD_KEYS:
            continue
        d = np.abs(p.detach().cpu().numpy() - tr2.P[k].numpy())        # Adam step 1 = lr * g / (|g| + eps): round-off-sized g may flip
        assert d.mean() < 6e-6 and d.max() <= 6.1e-4, (k, d.mean(), d.max())
    sd = m.state_dict()
    for k in ("enc_eeg.0.tsconv.2.running_mean", "enc_eeg.0.tsconv.2.running_var", "enc_eeg.0.tsconv.5.running_mean", "enc_eeg.0.tsconv.5.running_var"):
        np.testing.assert_allclose(sd[k].cpu().numpy(), tr2.P[k].numpy(), atol=2e-5, err_msg=k)


def test_step_plan_at_batch_256_matches_the_oracle_step_by_step():
    """the path bench.py times.  Five retrieval.contrastive_step calls at B = 256 (dropout off): steps 1 - 3 are the ordinary warm-up, steps 4 - 5 run
    as ONE eegclip_plan_run each (step_plan.StepPlan).  EVERY step is held against oracle.OracleTrainer.step at the single-step tolerances of
    test_train_step_at_batch_256_matches_the_oracle: before each step the oracle takes over the model's current state (parameters, BatchNorm buffers, AdamW
    moments and step count), so that what is compared is that step's own arithmetic -- loss, embeddings, running accuracy, post-AdamW parameters, BatchNorm
    running statistics -- not five steps of AdamW-amplified round-off."""
    from eeg_image_decode_amd import optim, retrieval, step_plan
    from oracle import loss as oloss
    state_np = syn.make_state(SEED, oatms.state_spec())
    B, NC, steps = 256, 200, 5
    cls = T(syn.unit_features(SEED + 4, NC, tag="c"))
    rng = np.random.default_rng(1)
    data = [(T(syn.eeg_batch(SEED + 100 + i, B)), T(syn.unit_features(SEED + 200 + i, B, tag="i")), T(syn.unit_features(SEED + 300 + i, B, tag="t")),
             T(rng.integers(0, NC, size=B).astype(np.int64))) for i in range(steps)]
    m = make_model(state_np)
    zero_dropout(m)
    m.train()
    opt = optim.AdamW(m.parameters(), lr=3e-4)
    tr = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    names = [k for k, _ in m.named_parameters()]
    acc, correct = [], torch.zeros(1, dtype=torch.int32, device="cuda")
    cg = cls.cuda()
    for i, (x, img, txt, lab) in enumerate(data):
        # the oracle starts this step from the model's state
        sd = m.state_dict()
        for k in tr.P:
            tr.P[k] = sd[k].detach().cpu().clone()
        st = opt.state_dict()["state"]
        tr.t = i
        for idx, k in enumerate(names):
            if idx in st:
                assert int(st[idx]["step"]) == i
                tr.m[k], tr.v[k] = st[idx]["exp_avg"].cpu().numpy().copy(), st[idx]["exp_avg_sq"].cpu().numpy().copy()
        before = int(correct)
        z = retrieval.contrastive_step(m, opt, x.cuda(), 1, img.cuda(), txt.cuda(), lab.cuda(), cg, acc, correct)
        on_plan = bool(retrieval.step_plans_of(m))
        assert on_plan == (i >= step_plan.StepPlan.WARM_STEPS), (i, on_plan)          # (an invalidated plan would have been dropped from the table)
        scale_before = tr.P["logit_scale"].clone()
        lo, zo = tr.step(x, torch.full((B,), 1).long(), img, txt)
        np.testing.assert_allclose(z.cpu().numpy(), zo.numpy(), atol=1e-4, err_msg=f"embeddings, step {i}")
        assert abs(float(acc[-1]) - float(lo)) < 1e-4, (i, float(acc[-1]), float(lo))
        want = int((oloss.train_accuracy_predictions(zo, cls, scale_before) == lab).sum())
        assert abs(int(correct) - before - want) <= 1, (i, int(correct) - before, want)         # (one near-tie may rank the other way)
        for k, p in m.named_parameters():
            if p.grad is None and k not in tr.m:
                continue
            if k in oloops.ZERO_GRAD_KEYS:
                continue
            d = np.abs(p.detach().cpu().numpy() - tr.P[k].numpy())        # an element whose gradient is round-off-sized may move by up to ~lr either way
            assert d.mean() < 6e-6 and d.max() <= 6.1e-4, (i, k, float(d.mean()), float(d.max()))
        sd = m.state_dict()
        for k in ("enc_eeg.0.tsconv.2.running_mean", "enc_eeg.0.tsconv.2.running_var", "enc_eeg.0.tsconv.5.running_mean", "enc_eeg.0.tsconv.5.running_var"):
            np.testing.assert_allclose(sd[k].cpu().numpy(), tr.P[k].numpy(), atol=2e-5, err_msg=f"{k}, step {i}")
        assert int(sd["enc_eeg.0.tsconv.2.num_batches_tracked"]) == i + 1
    assert all(p.grad is None for p in m.parameters()) and float(m._engine().gflat.abs().max()) == 0.0
    assert opt.state_dict()["state"][0]["step"] == steps


def test_train_step_at_batch_256_with_real_dropout_matches_the_oracle_under_the_same_philox_masks():
    state_np = syn.make_state(SEED, oatms.state_spec())
    B = 256
    x = T(syn.eeg_batch(SEED + 71, B))
    img, txt = T(syn.unit_features(SEED + 71, B, tag="img")), T(syn.unit_features(SEED + 71, B, tag="txt"))
    m = make_model(state_np).train()
    z = m(x.cuda(), 1)
    loss = m.loss_func.forward_mixed(z, [(img.cuda(), 0.99), (txt.cuda(), 0.01)], m.logit_scale)
    loss.backward()
    seed = m._engine().bufs[B]["seed"]
    shapes = {"embed": (B, 64, 250), "attn": (B, 4, 64, 64), "attn_out": (B, 64, 250), "ffn_act": (B, 64, 256), "ffn_out": (B, 64, 250),
              "conv": (B, 40, 1, 36), "proj": (B, 1024)}
    ps = {"embed": .25, "attn": .25, "attn_out": .25, "ffn_act": .25, "ffn_out": .25, "conv": .5, "proj": .5}
    masks = {s: T(keep_mask(seed, i, int(np.prod(shapes[s])), ps[s]).reshape(shapes[s])) for i, s in enumerate(oatms.DROPOUT_SITES)}
    tr = oloops.OracleTrainer(oloops.torch_state(state_np))
    lo, zo, grads, _ = tr.loss_and_grads(x, torch.full((B,), 1).long(), img, txt, train=True, masks=masks)
    np.testing.assert_allclose(z.detach().cpu().numpy(), zo.numpy(), atol=2e-4)
    assert abs(float(loss) - float(lo)) < 2e-4
    _check_grads(m, {k: (v.numpy() if v is not None else None) for k, v in grads.items()}, atol_rel=4e-3)


# ------------------------------------------------------------------------------------------------------------ configs[2]
def _loss_worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    n = 8
    a_all = T(syn.unit_features(SEED + 7, n * world, tag="da") * 32.0)
    b_all = T(syn.unit_features(SEED + 7, n * world, tag="db"))
    a = a_all[rank * n:(rank + 1) * n].clone().cuda().requires_grad_(True)
    b = b_all[rank * n:(rank + 1) * n].clone().cuda().requires_grad_(True)
    from eeg_image_decode_amd.loss import ClipLoss
    loss = ClipLoss(local_loss=mode[0], gather_with_grad=mode[1], rank=rank, world_size=world)(a, b, torch.tensor(float(np.log(1 / 0.07)), device="cuda"))
    loss.backward()
    torch.cuda.synchronize()
    ret[rank] = (float(loss), a.grad.cpu().numpy()[:, :128].copy(), b.grad.cpu().numpy()[:, :128].copy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode", [(False, False), (False, True), (True, True)])
def test_clip_loss_gather_modes_on_the_gpu_match_the_reference_gloo_fixture(world, mode):
    """B2 (models/loss.py:20-75): per-rank loss and per-rank feature gradients of all three gather modes, HIP kernels on every rank"""
    g = np.load(os.path.join(GOLDEN, "dist_loss.npz"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_loss_worker, args=(world, 29771 + 8 * world + 2 * int(mode[0]) + int(mode[1]), mode, ret), nprocs=world, join=True)
    tag = f"w{world}_ll{int(mode[0])}_gwg{int(mode[1])}"
    for r in range(world):
        assert abs(ret[r][0] - g[tag + "_loss"][r]) < 2e-5
        np.testing.assert_allclose(ret[r][1], g[tag + "_da"][r], atol=3e-6)
        np.testing.assert_allclose(ret[r][2], g[tag + "_db"][r], atol=6e-5)


# ------------------------------------------------------------------------------------------------------------ configs[3]
def _prior(dropout=0.0):
    from eeg_image_decode_amd.prior import DiffusionPriorUNet
    m = DiffusionPriorUNet(cond_dim=1024, dropout=dropout)
    state = syn.make_state(SEED + 20, oprior.prior_state_spec())
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    return m.cuda(), oloops.torch_state(state)


def _prior_batch(Bn, seed):
    h = T(syn.unit_features(seed, Bn, tag="ph") * 6.0)
    cc = T(syn.unit_features(seed, Bn, tag="pcc") * 32.0)
    noise = T(syn.eeg_batch(seed, Bn, 1, 1024)[:, 0])
    ts = torch.from_numpy(np.random.default_rng(seed).integers(0, 1000, Bn))
    return h, cc, noise, ts


def test_prior_objective_and_every_gradient_at_batch_1024_match_the_oracle():
    """E1/E2 at the notebook's batch size (diffusion_prior.py:167-203, 314-325): epsilon prediction, loss and all parameter gradients"""
    from eeg_image_decode_amd.prior import DDPMScheduler
    m, P = _prior()
    m.train()
    h, cc, noise, ts = _prior_batch(1024, SEED + 80)
    pert = DDPMScheduler().add_noise(h.cuda(), noise.cuda(), ts.cuda())
    pred = m(pert, ts.cuda(), cc.cuda())
    loss = ((pred - noise.cuda()) ** 2).mean()
    loss.backward()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lo, po = oprior.prior_loss(Pg, h, noise, ts, cc, oprior.DDPMSchedulerOracle())
    lo.backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), po.detach().numpy(), atol=3e-4)
    assert abs(float(loss) - float(lo)) < 1e-4
    for k, p in m.named_parameters():
        r = Pg[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, atol=1e-7 + 3e-3 * np.abs(r).max(), err_msg=k)
    # unconditional branch (the 10 % whole-batch drop): the condition layers get no gradient at all
    m.zero_grad(set_to_none=True)
    pred_u = m(pert, ts.cuda())
    ((pred_u - noise.cuda()) ** 2).mean().backward()
    pu = oprior.prior_unet_forward(P, oprior.DDPMSchedulerOracle().add_noise(h, noise, ts), ts, None)
    np.testing.assert_allclose(pred_u.detach().cpu().numpy(), pu.numpy(), atol=3e-4)
    assert all(p.grad is None for k, p in m.named_parameters() if "cond_embedding" in k)


def _inject_rng(pp, noises, tss):
    """make Pipe.train consume a given (noise, timesteps) sequence instead of torch's generators"""
    it_n, it_t = iter(noises), iter(tss)
    pp.torch.randn_like = lambda t_: next(it_n).to(t_.device)
    pp.torch.randint = lambda lo, hi, shape, device=None: next(it_t).to(device)


def _draw_like_the_reference(seed, batches, epochs):
    """torch's global CPU stream in Pipe.train's order (rand(1), randn_like(h), randint(0, 1000, (N,))): returns (drops, noises, timesteps)"""
    torch.manual_seed(seed)
    drops, noises, tss = [], [], []
    for _ in range(epochs):
        for bt in batches:
            drops.append(bool(torch.rand(1) < 0.1))
            noises.append(torch.randn_like(bt["h_embedding"]))
            tss.append(torch.randint(0, 1000, (bt["h_embedding"].shape[0],)))
    return drops, noises, tss


def _seed_without_drops(batches, epochs):
    for seed in range(1000, 1100):
        if not any(_draw_like_the_reference(seed, batches, epochs)[0]):
            return seed
    raise AssertionError("no seed found")


def test_pipe_train_steps_at_batch_1024_match_the_oracle():
    """E2 control flow at batch 1024: add_noise, epsilon MSE, backward, global-norm clip at 1.0, scheduler step BEFORE the Adam step"""
    import eeg_image_decode_amd.prior as pp
    h, cc, _, _ = _prior_batch(2048, SEED + 81)
    data = [{"c_embedding": cc[:1024], "h_embedding": h[:1024]}, {"c_embedding": cc[1024:], "h_embedding": h[1024:]}]
    seed = _seed_without_drops(data, 1)
    _, noises, tss = _draw_like_the_reference(seed, data, 1)
    m, P0 = _prior()
    pipe = pp.Pipe(m, device="cuda")
    pipe.cond_drop_prob = 0.0
    real = (torch.randn_like, torch.randint)
    try:
        _inject_rng(pp, noises, tss)
        pipe.train(data, num_epochs=1, learning_rate=1e-3)
    finally:
        pp.torch.randn_like, pp.torch.randint = real
    torch.manual_seed(seed)
    Po, _, lrs = oprior.pipe_train(P0, data, 1, 1e-3)
    assert pipe.lr_history == lrs
    for k, p in m.named_parameters():
        d = np.abs(p.detach().cpu().numpy() - Po[k].numpy())
        assert d.mean() < 5e-7 and d.max() <= 1.3e-5, (k, d.mean(), d.max())       # two Adam steps at lr 2e-6 / 4e-6: |update| <= lr each


def _prior_dp_worker(rank, world, port, seed, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(rank), "0", str(world)
    os.environ["EEGCLIP_DIST_BACKEND"] = "gloo"
    import eeg_image_decode_amd.prior as pp
    from eeg_image_decode_amd import dist as edist
    edist.init_from_env()
    h, cc, _, _ = _prior_batch(2048, SEED + 81)
    data = [{"c_embedding": cc[:1024], "h_embedding": h[:1024]}, {"c_embedding": cc[1024:], "h_embedding": h[1024:]}]
    _, noises, tss = _draw_like_the_reference(seed, data, 1)
    n = 1024 // world
    sl = slice(rank * n, (rank + 1) * n)
    mine = [{k: v[sl] for k, v in bt.items()} for bt in data]
    m, _ = _prior()
    pipe = pp.Pipe(m, device="cuda")
    pipe.cond_drop_prob = 0.0
    _inject_rng(pp, [z[sl] for z in noises], [t[sl] for t in tss])
    pipe.train(mine, num_epochs=1, learning_rate=1e-3)
    torch.cuda.synchronize()
    ret[rank] = {k: p.detach().cpu().numpy() for k, p in m.named_parameters()}
    dist.destroy_process_group()


def test_two_rank_prior_training_equals_the_single_process_batch_1024_steps():
    """configs[3] is data parallel: two ranks x 512 rows with the flat gradient averaged == one process x 1024 rows (the oracle)"""
    h, cc, _, _ = _prior_batch(2048, SEED + 81)
    data = [{"c_embedding": cc[:1024], "h_embedding": h[:1024]}, {"c_embedding": cc[1024:], "h_embedding": h[1024:]}]
    seed = _seed_without_drops(data, 1)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_prior_dp_worker, args=(2, 29791, seed, ret), nprocs=2, join=True)
    _, P0 = _prior()
    torch.manual_seed(seed)
    Po, _, _ = oprior.pipe_train(P0, data, 1, 1e-3)
    for k in ret[0]:
        np.testing.assert_array_equal(ret[0][k], ret[1][k], err_msg=k)
        d = np.abs(ret[0][k] - Po[k].numpy())
        assert d.mean() < 5e-7 and d.max() <= 1.3e-5, (k, d.mean(), d.max())


def test_pipe_train_lr_sequence_over_650_updates_including_the_cosine_branch():
    """E2: 500 warm-up updates, then the cosine branch; the scheduler is stepped before the optimizer (diffusion_prior.py:331-332), so update k
    runs at factor(k).  Expected values from transformers' get_cosine_schedule_with_warmup -- the function diffusers' is a copy of."""
    from transformers.optimization import get_cosine_schedule_with_warmup
    import eeg_image_decode_amd.prior as pp
    rng = np.random.default_rng(3)
    data = [{"c_embedding": T(rng.standard_normal((8, 1024)).astype(np.float32)), "h_embedding": T(rng.standard_normal((8, 1024)).astype(np.float32))}
            for _ in range(50)]
    m, _ = _prior()
    pipe = pp.Pipe(m, device="cuda")
    pipe.train(data, num_epochs=13, learning_rate=1e-3)
    dummy = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sch = get_cosine_schedule_with_warmup(dummy, num_warmup_steps=500, num_training_steps=650)
    want = []
    for _ in range(650):
        sch.step()
        want.append(dummy.param_groups[0]["lr"])
        dummy.step()
    assert len(pipe.lr_history) == 650
    np.testing.assert_allclose(pipe.lr_history, want, rtol=1e-12, atol=1e-18)
    assert want[499] == 1e-3 and 0 < want[-2] < 1e-5 and want[-1] < 1e-12
    assert all(torch.isfinite(p).all() for p in m.parameters())


# ------------------------------------------------------------------------------------------------------------ configs[4]
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hw,dim,heads", [(4096, 640, 10), (1024, 1280, 20)])
def test_cross_attention_at_the_sdxl_sampling_shapes(dtype, hw, dim, heads):
    """8 images per GPU x the classifier-free-guidance pair, 77 text + 4 image tokens: the full-workgroup (512 queries) instantiation of the
    kernel.  Query rows are independent, so the fp64 oracle scores a random subset of rows of every (sample, head) exactly; the remaining
    rows are covered by feeding the same queries at different positions (a permutation of the rows permutes the output)."""
    from eeg_image_decode_amd.sdxl import cross_attention
    torch.manual_seed(hw + dim)
    B = 16
    q = torch.randn(B, hw, dim, device="cuda", dtype=dtype)
    k, v = torch.randn(B, 77, dim, device="cuda", dtype=dtype), torch.randn(B, 77, dim, device="cuda", dtype=dtype)
    kip, vip = torch.randn(B, 4, dim, device="cuda", dtype=dtype), torch.randn(B, 4, dim, device="cuda", dtype=dtype)
    out = cross_attention(q, k, v, heads, kip, vip, 1.0)
    assert out.shape == q.shape and out.dtype == dtype and torch.isfinite(out).all()
    rows = np.sort(np.random.default_rng(1).choice(hw, 96, replace=False))
    f = lambda t: t.float().cpu().numpy()
    ref = sdxl_attn.cross_attention(f(q[:, rows]), f(k), f(v), heads, f(kip), f(vip), 1.0)
    tol = 6e-3 if dtype == torch.float16 else 2.5e-2
    np.testing.assert_allclose(f(out[:, rows]), ref, atol=tol)
    perm = torch.randperm(hw, device="cuda")
    out_p = cross_attention(q[:, perm].contiguous(), k, v, heads, kip, vip, 1.0)
    assert torch.equal(out_p, out[:, perm])                                          # every row, bit for bit, wherever it sits in the grid
    no_ip = cross_attention(q, k, v, heads)                                          # text branch alone (no IP-Adapter loaded)
    ref0 = sdxl_attn.cross_attention(f(q[:, rows]), f(k), f(v), heads)
    np.testing.assert_allclose(f(no_ip[:, rows]), ref0, atol=tol)


def test_kernel_timestamp_timing_of_one_launch():
    """eegclip_time_next_launch stamps the next kernel with its own GPU begin / end (hipExtLaunchKernel start / stop events): the figure bench.py's
    roofline uses.  It must be positive, below a record-around-the-launch bracket of the same launch, and a second launch without arming must
    leave the events untouched."""
    import ctypes
    from eeg_image_decode_amd import _abi
    from eeg_image_decode_amd._lib import lib
    L = lib()
    D = _abi.dim
    M, N, K = 16384, 256, 250
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    c = torch.empty(M, N, device="cuda")
    d = _abi.GemmDesc(M=M, N=N, K=K, A=a.data_ptr(), Am=D(K), Ak=D(1), B=w.data_ptr(), Bk=D(1), Bn=D(K), C=c.data_ptr(), Cm=D(N), Cn=D(1), Cpre=None,
                      bias_n=None, bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=0, act=0, drop_p=0.0, seed=0, drop_site=0, split_k=1,
                      precision=_abi.PREC_BF16X3)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert L.eegclip_gemm_f32(ctypes.byref(d), st) == 0
    e0, e1 = L.eegclip_timing_event_create(), L.eegclip_timing_event_create()
    assert e0 and e1
    assert L.eegclip_time_next_launch(e0, None) < 0                 # both or neither
    torch.cuda.synchronize()
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    assert L.eegclip_time_next_launch(e0, e1) == 0
    b0.record()
    assert L.eegclip_gemm_f32(ctypes.byref(d), st) == 0            # the stamped launch ITSELF inside an event bracket
    b1.record()
    assert L.eegclip_gemm_f32(ctypes.byref(d), st) == 0            # not armed: an ordinary launch (must leave e0 / e1 alone)
    torch.cuda.synchronize()
    ms = float(L.eegclip_timing_elapsed_ms(e0, e1))
    assert 0.003 < ms < 0.2, ms                                     # ~17 us for this shape
    # the kernel's own begin .. end lies INSIDE the bracket around the same launch (the bracket adds marker packets and dispatch gaps, never removes time)
    assert ms <= b0.elapsed_time(b1) + 0.0005, (ms, b0.elapsed_time(b1))
    np.testing.assert_allclose(c[:4].cpu().numpy(), (a[:4].double() @ w.double().T).cpu().numpy(), atol=2e-3)
    assert L.eegclip_timing_event_destroy(e0) == 0 and L.eegclip_timing_event_destroy(e1) == 0
