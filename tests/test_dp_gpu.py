"""The data-parallel step with the REAL HIP kernels and two processes: both ranks share the one GPU of the test box through the gloo backend
(RCCL refuses duplicate devices), so everything but the transport is the production path -- all-gathered negatives + reduce-scattered
embedding gradients in ClipLoss, SyncBN statistics all-reduced from inside the launch plans, the early gradient bucket reduced asynchronously
from the backward plan's second stream, the remainder after it.  Result must equal ONE process stepping on the global batch (the oracle)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import SEED
from eeg_image_decode_amd import synthetic as syn
from oracle import atms as oatms
from oracle import loops as oloops

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(rank), str(rank), str(world)
    os.environ["EEGCLIP_DIST_BACKEND"] = "gloo"
    from eeg_image_decode_amd import dist as edist
    from eeg_image_decode_amd import optim, retrieval
    from eeg_image_decode_amd.atms import ATMS
    edist.init_from_env()
    state_np = syn.make_state(SEED, oatms.state_spec())
    x_all = T(syn.eeg_batch(SEED + 41, n * world)).cuda()
    img_all = T(syn.unit_features(SEED + 41, n * world, tag="img")).cuda()
    txt_all = T(syn.unit_features(SEED + 41, n * world, tag="txt")).cuda()
    sl = slice(rank * n, (rank + 1) * n)
    m = ATMS()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state_np.items()})
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m = m.cuda().train()
    edist.configure_loss_for_world(m.loss_func, rank, world)
    opt = optim.AdamW(m.parameters(), lr=3e-4)
    loss_acc, correct = torch.zeros((), device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    classes = T(syn.unit_features(SEED + 42, 7, tag="cls")).cuda()
    calls = []                                                           # every collective of the step: (kind, payload bytes)
    saved = {}
    for kind in ("all_reduce", "all_gather_into_tensor", "reduce_scatter_tensor", "all_gather", "broadcast"):
        saved[kind] = getattr(dist, kind)

        def wrap(real, kind=kind):
            def call(*a, **k):
                t = a[0] if torch.is_tensor(a[0]) else a[0][0]
                calls.append((kind, t.numel() * t.element_size()))
                return real(*a, **k)
            return call
        setattr(dist, kind, wrap(saved[kind]))
    retrieval.contrastive_step(m, opt, x_all[sl].contiguous(), 1, img_all[sl].contiguous(), txt_all[sl].contiguous(),
                               torch.zeros(n, dtype=torch.long, device="cuda"), classes, loss_acc, correct)
    torch.cuda.synchronize()
    for kind, real in saved.items():
        setattr(dist, kind, real)
    eng = m._engine()
    early = any("allreduce_early_bucket" in pl.op_names() for k, pl in eng.plans.items() if k[0] == "b")
    ret[rank] = ({k: p.detach().cpu().numpy() for k, p in m.named_parameters()}, float(loss_acc), early,
                 {k: v.cpu().numpy() for k, v in m.state_dict().items() if "running" in k}, calls)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 4), (8, 32)])
def test_ranks_on_one_gpu_equal_the_single_process_global_batch_step(world, n):
    """(8, 32): the WHOLE step of a world-8 job -- configs[2]'s rank count -- with every rank's real kernels: SyncBN chain, early gradient bucket, the
    9 collectives of DESIGN.md section 7 with their payloads, rank-identical parameters, and the single-process global-batch step of the oracle"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29747 + world, n, ret), nprocs=world, join=True)
    state_np = syn.make_state(SEED, oatms.state_spec())
    x_all = T(syn.eeg_batch(SEED + 41, n * world))
    img_all, txt_all = T(syn.unit_features(SEED + 41, n * world, tag="img")), T(syn.unit_features(SEED + 41, n * world, tag="txt"))
    tr = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    lo, _ = tr.step(x_all, torch.full((n * world,), 1).long(), img_all, txt_all)
    p0, l0, e0, bn0, calls0 = ret[0]
    assert all(ret[r][2] for r in range(world))                          # the asynchronous early-bucket route was taken on every rank
    assert abs(float(np.mean([ret[r][1] for r in range(world)])) - float(lo)) < 1e-4
    # the step's collectives, identical on every rank: targets gathered in ONE call, Z gathered, its gradient reduce-scattered,
    # four SyncBN sums of 80 doubles, the flat gradient in two all-reduces (early bucket + rest)
    D_, nP = 1024, sum(int(np.prod(v.shape)) for v in p0.values())
    kinds = sorted(c[0] for c in calls0)
    assert kinds == sorted(["all_gather_into_tensor"] * 2 + ["reduce_scatter_tensor"] + ["all_reduce"] * 6), calls0
    assert all(ret[r][4] == calls0 for r in range(world))
    by = {}
    for kind, nbytes in calls0:
        by.setdefault(kind, []).append(nbytes)
    assert sorted(by["all_gather_into_tensor"]) == sorted([world * n * D_ * 4, 2 * world * n * D_ * 4])        # Z | [img | txt] targets, full gathered tensors
    assert by["reduce_scatter_tensor"] == [n * D_ * 4] or by["reduce_scatter_tensor"] == [world * n * D_ * 4]
    assert sorted(by["all_reduce"])[:4] == [640] * 4                     # SyncBN: 80 doubles, forward x 2 + backward x 2
    assert 4 * nP <= sum(sorted(by["all_reduce"])[4:]) <= 4 * nP + 4096  # the flat gradient, once, in two buckets
    for r in range(1, world):
        for k in p0:
            np.testing.assert_array_equal(p0[k], ret[r][0][k], err_msg=f"rank {r}: {k}")          # ranks stay bit-identical
    for k in p0:
        if k in oloops.ZERO_GRAD_KEYS or tr.P[k].shape != p0[k].shape:
            continue
        d = np.abs(p0[k] - tr.P[k].numpy())
        assert d.mean() < 6e-6 and d.max() <= 6.1e-4, (k, d.mean(), d.max())       # Adam step 1 = lr * sign(g): round-off-sized g may flip
    for k in bn0:
        np.testing.assert_allclose(bn0[k], tr.P[k].numpy(), atol=2e-5, err_msg=k)


def _plan_worker(rank, world, port, n, steps, plan_on, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(rank), str(rank), str(world)
    os.environ["EEGCLIP_DIST_BACKEND"] = "gloo"
    os.environ["EEGCLIP_STEP_PLAN_DP"] = "1" if plan_on else "0"
    from eeg_image_decode_amd import dist as edist
    from eeg_image_decode_amd import optim, retrieval, step_plan
    from eeg_image_decode_amd.atms import ATMS
    edist.init_from_env()
    state_np = syn.make_state(SEED, oatms.state_spec())
    m = ATMS()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state_np.items()})
    m = m.cuda().train()                                                 # dropout ON: the plan must consume the RNG like the loop
    torch.manual_seed(100 + rank)
    edist.configure_loss_for_world(m.loss_func, rank, world)
    opt = optim.AdamW(m.parameters(), lr=3e-4)
    NC = 50
    classes = T(syn.unit_features(SEED + 42, NC, tag="cls")).cuda()
    acc, correct = [], torch.zeros(1, dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(7)
    on_plan = []
    for i in range(steps):
        x_all = T(syn.eeg_batch(SEED + 60 + i, n * world)).cuda()
        img_all, txt_all = T(syn.unit_features(SEED + 60 + i, n * world, tag="img")).cuda(), T(syn.unit_features(SEED + 60 + i, n * world, tag="txt")).cuda()
        lab = T(rng.integers(0, NC, size=n * world).astype(np.int64)).cuda()
        sl = slice(rank * n, (rank + 1) * n)
        retrieval.contrastive_step(m, opt, x_all[sl].contiguous(), 1, img_all[sl].contiguous(), txt_all[sl].contiguous(), lab[sl].contiguous(), classes, acc, correct)
        on_plan.append(bool(retrieval.step_plans_of(m)))
    torch.cuda.synchronize()
    plans = retrieval.step_plans_of(m)
    ret[rank] = ({k: p.detach().cpu().numpy() for k, p in m.named_parameters()}, [float(a) for a in acc], int(correct), on_plan,
                 [type(p).__name__ + ":" + str(getattr(p, "world", None)) for p in plans],
                 {k: v.cpu().numpy() for k, v in m.state_dict().items() if "running" in k})
    dist.destroy_process_group()


def test_data_parallel_step_plan_on_the_gpu_trains_like_the_launch_by_launch_data_parallel_loop():
    """step_plan.StepPlan(world = 2) with the REAL kernels (two gloo ranks sharing the GPU, n = 64 per rank, dropout on): 7 steps -- 3 ordinary warm-up steps,
    then the plan, whose C op array is cut into segments around the target all-gather, the data-parallel loss and the flat-gradient all-reduce -- against the
    same 7 steps with EEGCLIP_STEP_PLAN_DP=0.  Ranks stay bit-identical in both runs; losses, accuracy count, BatchNorm statistics and parameters agree
    between the runs as two runs of the launch-by-launch loop do (float atomics are unordered; AdamW's first updates are lr * sign(g))."""
    world, n, steps = 2, 64, 7
    runs = []
    for plan_on in (True, False):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_plan_worker, args=(world, 29761 + int(plan_on), n, steps, plan_on, ret), nprocs=world, join=True)
        runs.append({r: ret[r] for r in range(world)})
    a, b = runs
    assert a[0][3] == [False] * 3 + [True] * (steps - 3) and a[0][4] == ["StepPlan:2"], (a[0][3], a[0][4])       # the plan took over after the warm-up ...
    assert not any(b[0][3]) and b[0][4] == []                                                                     # ... and never existed in the reference run
    for run in (a, b):
        for k in run[0][0]:
            np.testing.assert_array_equal(run[0][0][k], run[1][0][k], err_msg=f"ranks differ: {k}")
    np.testing.assert_allclose(a[0][1], b[0][1], rtol=2e-4)
    assert abs(a[0][2] - b[0][2]) <= 2
    for k in a[0][5]:
        np.testing.assert_allclose(a[0][5][k], b[0][5][k], rtol=1e-3, atol=1e-5, err_msg=k)
    for k in a[0][0]:
        if k.endswith("key_projection.bias"):
            continue
        d = np.abs(a[0][0][k] - b[0][0][k])
        assert d.max() <= steps * 3e-4 * 1.01 and (d > 3e-4).mean() <= 0.02, (k, float(d.max()), float((d > 3e-4).mean()))


def test_bench_runs_on_two_ranks_and_prints_one_json_line():
    """`bench.py --gpus 2` the way the driver launches it (torch.distributed.run, one process per rank; here both ranks on the one GPU through
    gloo): every rank must issue the same number of steps -- a time-based warm-up loop once ran a different count on each rank and hung the
    collectives -- and rank 0 prints exactly one JSON line with the whole-job throughput."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EEGCLIP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-secondary", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["config"]["global_batch"] == 512 and d["value"] > 0 and d["scaling"] == "weak"
    # what the run was + where its communication goes (the first multi-GPU line of the driver must explain itself)
    ds = d["distributed"]
    assert ds["backend"] == "gloo" and ds["world_size"] == 2 and ds["loss_mode"] == {"local_loss": True, "gather_with_grad": True}
    c = ds["collectives"]
    assert c["per_step"] == 9 and c["sum_us_stream_held_per_step"] > 0, c       # 9 collectives per step
