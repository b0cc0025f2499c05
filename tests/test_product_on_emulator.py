"""Product host code + kernel sources together on CPU (lane emulator, see tests/emu_patch.py): full ATMS forward/backward plan,
ClipLoss autograd, fused AdamW, train_model, and the 2-rank gloo data-parallel step (all-gather negatives, reduce-scatter of the
embedding gradients, SyncBN, flat-gradient all-reduce) against the single-process oracle."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import SEED
from emu_patch import product_on_emulator
from eeg_image_decode_amd import synthetic as syn
from oracle import atms as oatms
from oracle import loops as oloops
from oracle import loss as oloss

pytestmark = pytest.mark.emu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_model(state_np):
    from eeg_image_decode_amd.atms import ATMS
    m = ATMS()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state_np.items()})
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def test_atms_train_step_under_emulator_matches_oracle():
    state_np = syn.make_state(SEED, oatms.state_spec())
    B = 3
    x = T(syn.eeg_batch(SEED + 40, B))
    img, txt = T(syn.unit_features(SEED + 40, B, tag="img")), T(syn.unit_features(SEED + 40, B, tag="txt"))
    with product_on_emulator():
        from eeg_image_decode_amd import optim
        m = make_model(state_np).train()
        opt = optim.AdamW(m.parameters(), lr=3e-4)
        z = m(x, 1)
        loss = 0.99 * m.loss_func(z, img, m.logit_scale) + 0.01 * m.loss_func(z, txt, m.logit_scale)
        loss.backward()
        grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}
        opt.step()
        after = {k: p.detach().clone() for k, p in m.named_parameters()}
    tr = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    lo, zo, og, _ = tr.loss_and_grads(x, torch.full((B,), 1).long(), img, txt, train=True)
    np.testing.assert_allclose(z.detach().numpy(), zo.numpy(), atol=1e-4)
    assert abs(float(loss) - float(lo)) < 1e-4
    for k, g in grads.items():
        if og[k] is None:
            assert g is None, k
        elif k not in oloops.ZERO_GRAD_KEYS:
            np.testing.assert_allclose(g.numpy(), og[k].numpy(), atol=1e-7 + 3e-3 * float(og[k].abs().max()), err_msg=k)
    tr2 = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    tr2.step(x, torch.full((B,), 1).long(), img, txt)
    for k in ("proj_eeg.0.weight", "encoder.encoder.attn_layers.0.attention.value_projection.weight", "logit_scale", "enc_eeg.0.tsconv.4.weight"):
        d = np.abs(after[k].numpy() - tr2.P[k].numpy())              # one fused AdamW step; the first Adam step is lr*sign(g), so
        assert d.mean() < 3e-6 and d.max() <= 6.1e-4, (k, d.mean(), d.max())   # elements with |g| ~ round-off may flip by up to 2*lr


def test_mixed_clip_loss_equals_the_two_reference_calls():
    """ClipLoss.forward_mixed (one pass, shared accumulator, accumulated dA GEMM) == alpha*L(z,img) + (1-alpha)*L(z,txt) -- value,
    d/dz, d/dscale and the gradient of a target that requires it"""
    rng = np.random.default_rng(3)
    n, Dm = 6, 16
    z0, img0, txt0 = [rng.standard_normal((n, Dm)).astype(np.float32) for _ in range(3)]
    with product_on_emulator():
        from eeg_image_decode_amd.loss import ClipLoss
        lf = ClipLoss()
        outs = []
        for mixed in (False, True):
            z, img, txt = T(z0).requires_grad_(), T(img0).requires_grad_(), T(txt0)
            sc = torch.tensor(2.6593, requires_grad=True)
            if mixed:
                loss = lf.forward_mixed(z, [(img, 0.99), (txt, 0.01)], sc)
            else:
                loss = 0.99 * lf(z, img, sc) + 0.01 * lf(z, txt, sc)
            (3.0 * loss).backward()
            outs.append([loss.detach().clone(), z.grad.clone(), img.grad.clone(), sc.grad.clone()])
    for a, b in zip(*outs):
        np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=2e-5, atol=1e-6)
    ref = oloss.mixed_loss(T(z0), T(img0), T(txt0), torch.tensor(2.6593)) if hasattr(oloss, "mixed_loss") else None
    if ref is not None:
        assert abs(float(outs[1][0]) - float(ref)) < 1e-5


def test_step_and_zero_grad_in_one_pass_is_the_plain_loop():
    """contrastive_step with optim.AdamW (update + zero_grad in one kernel, the engine skips its own clear) against the same steps with an optimizer
    that only offers step(): the same parameters after 3 steps, and the gradients are None / the flat buffer clear in between"""
    state_np = syn.make_state(SEED, oatms.state_spec())
    B = 3
    batches = [(T(syn.eeg_batch(SEED + 50 + i, B)), T(syn.unit_features(SEED + 50 + i, B, tag="img")), T(syn.unit_features(SEED + 50 + i, B, tag="txt")))
               for i in range(3)]
    labels = torch.arange(B)
    finals = []
    with product_on_emulator():
        from eeg_image_decode_amd import atms, optim, retrieval

        class PlainStep(optim.AdamW):
            supports_step_and_zero_grad = False

        atms._Engine.check_cleared = True              # every skipped clear of the flat gradient buffer is verified (raises otherwise)
        try:
            for cls, keep in ((optim.AdamW, False), (PlainStep, False), (optim.AdamW, True)):
                m = make_model(state_np).train()
                opt = cls(m.parameters(), lr=3e-4)
                acc = [] if cls is optim.AdamW else torch.zeros(())
                correct = torch.zeros(1, dtype=torch.int32)
                torch.manual_seed(5)
                for x, img, txt in batches:
                    retrieval.contrastive_step(m, opt, x, 1, img, txt, labels, img, acc, correct, keep_grads=keep)
                    if cls is optim.AdamW and not keep:
                        assert all(p.grad is None for p in m.parameters())
                        assert float(m._engine().gflat.abs().max()) == 0.0
                    else:                                  # the reference's behaviour: gradients stay until the next zero_grad()
                        assert m.logit_scale.grad is not None and m.proj_eeg[0].weight.grad is not None
                finals.append(({k: p.detach().clone() for k, p in m.named_parameters()}, float(retrieval.running_loss(acc))))
        finally:
            atms._Engine.check_cleared = False
    assert abs(finals[0][1] - finals[1][1]) < 1e-5 * abs(finals[1][1]) and abs(finals[2][1] - finals[1][1]) < 1e-5 * abs(finals[1][1])
    # (gradient sums through atomics are not bit-reproducible run to run, and Adam's first steps move an element by lr * g / |g|: round-off-sized
    # gradients differ by a fraction of lr = 3e-4 between ANY two runs; a stale or uncleared gradient would move most elements by ~lr)
    for k in finals[0][0]:
        if k.endswith("key_projection.bias"):          # its true gradient is zero (softmax shift invariance): pure round-off under Adam
            continue
        for other in (0, 2):
            d = np.abs(finals[other][0][k].numpy() - finals[1][0][k].numpy())
            assert d.max() <= 3 * 3e-4 * 1.01 and (d > 2e-5).mean() <= 2e-3, (k, float(d.max()), float((d > 2e-5).mean()))


def test_gradient_buffer_attach_accumulates_clears_and_keeps_foreign_gradients():
    """engine.attach_grads: a second backward without zero_grad accumulates (g1 + g2); after zero_grad the buffer starts from zero; a gradient that
    somebody else put on a parameter before the backward (a foreign tensor) is added to, not lost; a parameter whose .grad alone was set to None is
    cleared alone"""
    state_np = syn.make_state(SEED, oatms.state_spec())
    B = 2
    xs = [T(syn.eeg_batch(SEED + 70 + i, B)) for i in range(2)]
    img = T(syn.unit_features(SEED + 70, B, tag="img"))
    key = "encoder.encoder.attn_layers.0.conv1.weight"
    with product_on_emulator():
        m = make_model(state_np).eval()                       # eval: no dropout, deterministic BatchNorm -> the two backwards are independent
        named = dict(m.named_parameters())

        def backward(x):
            m.loss_func(m(x, 1), img, m.logit_scale).backward()

        def grads():
            return {k: p.grad.clone() for k, p in named.items() if p.grad is not None}

        backward(xs[0]); g1 = grads()
        m.zero_grad(set_to_none=True)
        backward(xs[1]); g2 = grads()
        backward(xs[0]); g21 = grads()                        # no zero_grad in between: accumulated
        for k in g1:
            np.testing.assert_allclose(g21[k].numpy(), (g1[k] + g2[k]).numpy(), atol=1e-6 + 1e-4 * float((g1[k].abs() + g2[k].abs()).max()), err_msg=k)
        m.zero_grad(set_to_none=True)
        named[key].grad = torch.full_like(named[key], 0.25)   # a foreign gradient tensor on one parameter
        backward(xs[0]); gf = grads()
        np.testing.assert_allclose(gf[key].numpy(), (g1[key] + 0.25).numpy(), atol=1e-6 + 1e-4 * float(g1[key].abs().max()))
        other = "proj_eeg.0.weight"
        np.testing.assert_allclose(gf[other].numpy(), g1[other].numpy(), atol=1e-6 + 1e-4 * float(g1[other].abs().max()))
        named[key].grad = None                                # only this one cleared: the others keep accumulating
        backward(xs[1]); gm = grads()
        np.testing.assert_allclose(gm[key].numpy(), g2[key].numpy(), atol=1e-6 + 1e-4 * float(g2[key].abs().max()))
        np.testing.assert_allclose(gm[other].numpy(), (g1[other] + g2[other]).numpy(), atol=1e-6 + 1e-4 * float((g1[other].abs() + g2[other].abs()).max()))


def test_optimizer_fast_path_is_torch_adamw_and_keeps_step_counts():
    """the steady-state path of optim.AdamW.step (same gradient tensors as the previous step: cached launches, lazily counted steps) against
    torch.optim.AdamW over 6 steps with a learning-rate change in between; state_dict() reports the true step count"""
    rng = np.random.default_rng(3)
    with product_on_emulator():
        from eeg_image_decode_amd import optim
        flat, gflat = torch.zeros(300), torch.zeros(300)
        shapes, off, ps, gs = [(7, 9), (40,), (11, 3)], 0, [], []
        for sh in shapes:
            n = int(np.prod(sh))
            flat[off:off + n] = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
            ps.append(torch.nn.Parameter(flat[off:off + n].view(sh)))
            gs.append(gflat[off:off + n].view(sh))
            off += (n + 3) // 4 * 4
        ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        opt, topt = optim.AdamW(ps, lr=3e-3, weight_decay=0.02), torch.optim.AdamW(ref, lr=3e-3, weight_decay=0.02)
        for it in range(6):
            if it == 4:
                for o in (opt, topt):
                    o.param_groups[0]["lr"] = 1e-3
            for p, g, r in zip(ps, gs, ref):
                g.copy_(torch.from_numpy(rng.standard_normal(tuple(g.shape)).astype(np.float32)))
                p.grad = g                                   # the SAME tensor objects every step, as the engine attaches them
                r.grad = g.clone()
            if it == 5:
                ps[1].grad = gs[1].clone()                   # another tensor object (not in the flat buffer): the cached launches must not be used
            opt.step(zero_grad=(it % 2 == 1))
            topt.step()
            assert (it < 2) or (it == 5) or opt._fast_last[0]["pending"] >= 1  # from the third step on the cached launches are used
            if it == 5:
                assert opt._fast_last[0]["pending"] == 0 and len(opt._fast_last[0]["launch"]) >= 2      # re-derived: the run is split at the foreign gradient
            if it == 4:
                # copy.deepcopy / pickle go through __getstate__, not state_dict(): the lazily counted steps must be flushed there too
                import copy
                assert opt._fast_last[0]["pending"] >= 1
                dup = copy.deepcopy(opt)
                assert [int(st["step"]) for st in dup.state.values()] == [5, 5, 5] and dup._fast == {} and dup._moments == {}
            if it % 2 == 1 and it != 5:
                assert all(p.grad is None for p in ps) and float(gflat.abs().max()) == 0.0
        for p, r in zip(ps, ref):
            np.testing.assert_allclose(p.detach().numpy(), r.detach().numpy(), atol=2e-6, rtol=1e-5)
        sd = opt.state_dict()["state"]
        assert [int(sd[i]["step"]) for i in range(3)] == [6, 6, 6]


def test_optimizer_alternating_live_sets_keep_their_cached_launches():
    """the joint-subject model's pattern (the reference's joint loop alternates subjects batch by batch, ATMS_retrieval_joint_train.py:219-222): the
    parameters with a gradient change from step to step between a few sets.  Every set keeps its cached launches; switching settles the lazily
    counted steps, and a run whose members' step counts have drifted apart is re-derived.  Against torch.optim.AdamW, step counts included."""
    rng = np.random.default_rng(8)
    with product_on_emulator():
        from eeg_image_decode_amd import optim
        flat, gflat = torch.zeros(400), torch.zeros(400)
        shapes, off, ps, gs = [(6, 5), (12,), (8, 4), (8,), (8, 4), (8,)], 0, [], []      # shared weight, shared bias, subject A (w, b), subject B (w, b)
        for sh in shapes:
            n = int(np.prod(sh))
            flat[off:off + n] = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
            ps.append(torch.nn.Parameter(flat[off:off + n].view(sh)))
            gs.append(gflat[off:off + n].view(sh))
            off += (n + 3) // 4 * 4
        ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        opt, topt = optim.AdamW(ps, lr=2e-3, weight_decay=0.01), torch.optim.AdamW(ref, lr=2e-3, weight_decay=0.01)
        sets = {"A": [0, 1, 2, 3], "B": [0, 1, 4, 5], "AB": [0, 1, 2, 3, 4, 5]}
        order = ["A", "B", "A", "A", "B", "AB", "B", "B", "A", "AB", "AB", "A"]
        fast_hits = 0
        for it, name in enumerate(order):
            for i, (p, g, r) in enumerate(zip(ps, gs, ref)):
                if i in sets[name]:
                    g.copy_(torch.from_numpy(rng.standard_normal(tuple(g.shape)).astype(np.float32)))
                    p.grad, r.grad = g, g.clone()
                else:
                    p.grad, r.grad = None, None
            before = opt._fast_last.get(0)
            known = tuple(map(id, [p.grad for p in ps])) in opt._fast.get(0, {})
            opt.step(zero_grad=True)
            topt.step()
            fast_hits += int(known and opt._fast_last[0]["pending"] >= 1)
            assert all(p.grad is None for p in ps)
        assert fast_hits >= 6                                  # every set after its first appearance, unless its runs had to be re-derived
        for p, r in zip(ps, ref):
            np.testing.assert_allclose(p.detach().numpy(), r.detach().numpy(), atol=2e-6, rtol=1e-5)
        sd, tsd = opt.state_dict()["state"], topt.state_dict()["state"]
        assert [int(sd[i]["step"]) for i in range(6)] == [int(tsd[i]["step"]) for i in range(6)] == [12, 12, 8, 8, 7, 7]


def test_reconstruction_objective_step_under_emulator_matches_oracle():
    """10 * (0.9 MSE + 0.1 image InfoNCE) through contrastive_step(objective="reconstruction"): loss and parameters after one AdamW step"""
    state_np = syn.make_state(SEED, oatms.state_spec())
    B = 3
    x = T(syn.eeg_batch(SEED + 41, B))
    img, txt = T(syn.unit_features(SEED + 41, B, tag="img")), T(syn.unit_features(SEED + 41, B, tag="txt"))
    labels = torch.arange(B)
    with product_on_emulator():
        from eeg_image_decode_amd import optim, retrieval
        m = make_model(state_np).train()
        opt = optim.AdamW(m.parameters(), lr=3e-4)
        loss_acc, correct = torch.zeros(()), torch.zeros(1, dtype=torch.int32)
        retrieval.contrastive_step(m, opt, x, 1, img, txt, labels, img, loss_acc, correct, alpha=0.90, objective="reconstruction")
        after = {k: p.detach().clone() for k, p in m.named_parameters()}
    tr = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0, objective="reconstruction")
    lo, _ = tr.step(x, torch.full((B,), 1).long(), img, txt)
    assert abs(float(loss_acc) - float(lo)) < 2e-4 * max(1.0, abs(float(lo)))
    for k in ("proj_eeg.0.weight", "encoder.encoder.attn_layers.0.attention.query_projection.weight", "enc_eeg.0.tsconv.0.weight", "logit_scale"):
        d = np.abs(after[k].numpy() - tr.P[k].numpy())
        # the first Adam step is lr * g / (|g| + eps): an element whose gradient is round-off-sized moves by up to +-lr in either direction
        assert d.max() <= 2 * 3e-4 + 1e-6 and (d > 2e-5).mean() < 1e-4, (k, float(d.max()), float((d > 2e-5).mean()))


@pytest.mark.parametrize("ids", [[3, 0, 3, 7], [0, 3, 3, 7], [5, 5, 5, 5]], ids=["mixed-unordered", "mixed-ordered", "uniform"])
def test_joint_subject_model_under_emulator_matches_oracle(ids):
    """one value embedding per subject (retrieval_joint.ATMS(joint_train=True)).  Batches that mix subjects (gathered into
    subject order / already ordered) and the reference loops' uniform-id case: embeddings, loss, every parameter gradient (None for the
    subjects absent from the batch, as in the reference) and the input gradient."""
    state_np = syn.make_state(SEED, oatms.state_spec(joint_train=True, num_subjects=10))
    B = len(ids)
    x0 = syn.eeg_batch(SEED + 42, B)
    img, txt = T(syn.unit_features(SEED + 42, B, tag="img")), T(syn.unit_features(SEED + 42, B, tag="txt"))
    idt = torch.tensor(ids)
    with product_on_emulator():
        from eeg_image_decode_amd.retrieval_joint import ATMS
        m = ATMS(joint_train=True)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state_np.items()})
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        with torch.no_grad():
            z_eval = m.eval()(T(x0), idt).clone()         # before the train-mode forward moves the BatchNorm running statistics
        m.train()
        x = T(x0).requires_grad_()
        z = m(x, idt)
        loss = 0.99 * m.loss_func(z, img, m.logit_scale) + 0.01 * m.loss_func(z, txt, m.logit_scale)
        loss.backward()
        grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}
        dx = x.grad.clone()
        # the joint-subject plans run the fused transformer block (per-sample value-embedding base) and take the per-subject weight gradients from the
        # token planes through the subject-ordered sample list; only the input gradient (not part of training) still uses the grouped GEMM
        eng = m._engine()
        fnames = eng.plans[next(k for k in eng.plans if k[0] == "f" and k[2])].op_names()
        assert "eegclip_token_block_fwd" in fnames and "eegclip_token_block_pack_embed" in fnames and "eegclip_gemm_f32_grouped" not in fnames
        bnames = eng.plans[next(k for k in eng.plans if k[0] == "b")].op_names()
        assert bnames.count("eegclip_wgrad_tok") == 2 and bnames.count("eegclip_gemm_f32_grouped") == 1
        # a training step proper (no input gradient): no grouped GEMM, no subject-ordered copies at all
        for p_ in m.parameters():
            p_.grad = None
        z2 = m(T(x0), idt)
        (0.99 * m.loss_func(z2, img, m.logit_scale) + 0.01 * m.loss_func(z2, txt, m.logit_scale)).backward()
        bn2 = eng.plans[next(k for k in eng.plans if k[0] == "b" and not k[5])].op_names()
        assert "eegclip_gemm_f32_grouped" not in bn2 and "eegclip_gather_rows" not in bn2 and bn2.count("eegclip_wgrad_tok") == 2
        for k, p_ in m.named_parameters():
            if grads[k] is None:
                assert p_.grad is None, k
            else:
                np.testing.assert_allclose(p_.grad.numpy(), grads[k].numpy(), atol=1e-6 + 1e-4 * float(grads[k].abs().max()), err_msg=k)
        with pytest.raises(Exception, match="value embeddings for subjects"):
            m(T(x0), torch.tensor([0, 1, 2, 10][:B]))
    tr = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    lo, zo, og, _ = tr.loss_and_grads(T(x0), idt, img, txt, train=True)
    xo = T(x0).requires_grad_()
    oloss.mixed_loss(oatms.atms_forward(tr.P, xo, idt, train=True, p_scale=0.0), img, txt, tr.P["logit_scale"]).backward()
    np.testing.assert_allclose(z.detach().numpy(), zo.detach().numpy(), atol=1e-4)
    assert abs(float(loss.detach()) - float(lo)) < 1e-4
    present = set(ids)
    for k, g in grads.items():
        if og[k] is None:
            assert g is None, k
        elif k not in oloops.ZERO_GRAD_KEYS:
            assert g is not None, k
            np.testing.assert_allclose(g.numpy(), og[k].numpy(), atol=1e-7 + 3e-3 * float(og[k].abs().max()), err_msg=k)
    for s in range(10):
        assert (grads[f"encoder.enc_embedding.value_embedding.{s}.weight"] is not None) == (s in present)
    np.testing.assert_allclose(dx.numpy(), xo.grad.numpy(), atol=1e-7 + 3e-3 * float(xo.grad.abs().max()))
    zo_eval = oatms.atms_forward(tr.P, T(x0), idt, train=False)
    np.testing.assert_allclose(z_eval.numpy(), zo_eval.detach().numpy(), atol=1e-4)


def _dp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HIPEMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    state_np = syn.make_state(SEED, oatms.state_spec())
    n = 2
    x_all = T(syn.eeg_batch(SEED + 41, n * world))
    img_all, txt_all = T(syn.unit_features(SEED + 41, n * world, tag="img")), T(syn.unit_features(SEED + 41, n * world, tag="txt"))
    sl = slice(rank * n, (rank + 1) * n)
    with product_on_emulator():
        from eeg_image_decode_amd import dist as edist
        from eeg_image_decode_amd import optim, retrieval
        m = make_model(state_np).train()
        edist.configure_loss_for_world(m.loss_func, rank, world)
        opt = optim.AdamW(m.parameters(), lr=3e-4)
        loss_acc, correct = torch.zeros(()), torch.zeros(1, dtype=torch.int32)
        classes = T(syn.unit_features(SEED + 42, 7, tag="cls"))
        calls = {}

        def counted(name):
            fn = getattr(dist, name)

            def wrapper(*a, **k):
                calls[name] = calls.get(name, 0) + 1
                return fn(*a, **k)
            return wrapper
        saved = {nm: getattr(dist, nm) for nm in ("all_reduce", "all_gather_into_tensor", "reduce_scatter_tensor", "broadcast", "all_gather")}
        for nm in saved:
            setattr(dist, nm, counted(nm))
        try:
            retrieval.contrastive_step(m, opt, x_all[sl].contiguous(), 1, img_all[sl].contiguous(), txt_all[sl].contiguous(),
                                       torch.zeros(n, dtype=torch.long), classes, loss_acc, correct)
        finally:
            for nm, fn in saved.items():
                setattr(dist, nm, fn)
        # collectives of ONE data-parallel step: [img|txt] targets in one all-gather (started before the encoder), Z in one, one reduce-scatter of
        # the gathered-copy gradients, the flat gradient in two all-reduces (early bucket + rest), and the four data-dependent 640-byte SyncBN sums
        if os.environ.get("EEGCLIP_DP_OVERLAP", "1") != "0":
            assert calls == {"all_gather_into_tensor": 2, "reduce_scatter_tensor": 1, "all_reduce": 2 + 4}, calls
        eng = m._engine()
        # the conv-stack + head gradient bucket was all-reduced from INSIDE the backward plan (asynchronously) and collected afterwards
        early = [pl for k, pl in eng.plans.items() if k[0] == "b" and "allreduce_early_bucket" in pl.op_names()]
        assert bool(early) == (os.environ.get("EEGCLIP_DP_OVERLAP", "1") != "0") and eng.early_work is None
        ret[rank] = ({k: p.detach().clone().numpy() for k, p in m.named_parameters()}, float(loss_acc),
                     {k: v.clone().numpy() for k, v in m.state_dict().items() if "running" in k})
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_single_process_global_batch():
    """2 ranks x 2 samples with all-gathered negatives + SyncBN + averaged flat gradient  ==  1 process x 4 samples."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, 29733, ret), nprocs=world, join=True)
    state_np = syn.make_state(SEED, oatms.state_spec())
    x_all = T(syn.eeg_batch(SEED + 41, 4))
    img_all, txt_all = T(syn.unit_features(SEED + 41, 4, tag="img")), T(syn.unit_features(SEED + 41, 4, tag="txt"))
    tr = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    lo, _ = tr.step(x_all, torch.full((4,), 1).long(), img_all, txt_all)
    p0, l0, bn0 = ret[0]
    p1, l1, bn1 = ret[1]
    assert abs(0.5 * (l0 + l1) - float(lo)) < 1e-4            # local losses average to the global loss
    for k in p0:
        np.testing.assert_array_equal(p0[k], p1[k], err_msg=k)            # ranks stay bit-identical after the step
        if k in oloops.ZERO_GRAD_KEYS or tr.P[k].shape != p0[k].shape:
            continue
        d = np.abs(p0[k] - tr.P[k].numpy())                                # == single-process step on the global batch
        assert d.mean() < 6e-6 and d.max() <= 6.1e-4, (k, d.mean(), d.max())      # (Adam step 1 = lr*sign(g): round-off-sized g may flip)
    for k in bn0:
        np.testing.assert_allclose(bn0[k], tr.P[k].numpy(), atol=2e-5, err_msg=k)     # SyncBN running stats == global-batch stats


def _loss_worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 8
    a_all = T(syn.unit_features(SEED + 7, n * world, tag="da") * 32.0)
    b_all = T(syn.unit_features(SEED + 7, n * world, tag="db"))
    a = a_all[rank * n:(rank + 1) * n].clone().requires_grad_(True)
    b = b_all[rank * n:(rank + 1) * n].clone().requires_grad_(True)
    with product_on_emulator():
        from eeg_image_decode_amd.loss import ClipLoss
        l = ClipLoss(local_loss=mode[0], gather_with_grad=mode[1], rank=rank, world_size=world)(a, b, torch.tensor(float(np.log(1 / 0.07))))
        l.backward()
    ret[rank] = (float(l), a.grad.numpy()[:, :128].copy(), b.grad.numpy()[:, :128].copy())
    dist.destroy_process_group()


def _dp_plan_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HIPEMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    state_np = syn.make_state(SEED, oatms.state_spec())
    n, steps = 4, 4
    data = [(T(syn.eeg_batch(SEED + 50 + i, n * world)), T(syn.unit_features(SEED + 50 + i, n * world, tag="img")), T(syn.unit_features(SEED + 50 + i, n * world, tag="txt")))
            for i in range(steps)]
    sl = slice(rank * n, (rank + 1) * n)
    classes = T(syn.unit_features(SEED + 42, 7, tag="cls"))
    out = {}
    with product_on_emulator():
        from eeg_image_decode_amd import dist as edist
        from eeg_image_decode_amd import optim, retrieval, step_plan
        step_plan._runtime_ok = lambda: True
        step_plan._on_device = lambda t: True
        step_plan.StepPlan.WARM_STEPS = 2
        for mode in ("1", "0"):
            os.environ["EEGCLIP_STEP_PLAN"] = mode
            m = make_model(state_np).train()
            edist.configure_loss_for_world(m.loss_func, rank, world)
            opt = optim.AdamW(m.parameters(), lr=3e-4)
            acc, correct = [], torch.zeros(1, dtype=torch.int32)
            seen, ncalls = [], []
            real_ar = dist.all_reduce
            cnt = {"n": 0}

            def counted(*a, **k):
                cnt["n"] += 1
                return real_ar(*a, **k)
            dist.all_reduce = counted
            try:
                for x, img, txt in data:
                    cnt["n"] = 0
                    retrieval.contrastive_step(m, opt, x[sl].contiguous(), 1, img[sl].contiguous(), txt[sl].contiguous(), torch.zeros(n, dtype=torch.long), classes, acc,
                                               correct)
                    seen.append(bool(retrieval.step_plans_of(m)))
                    ncalls.append(cnt["n"])
            finally:
                dist.all_reduce = real_ar
            assert seen == ([False, False, True, True] if mode == "1" else [False] * 4), seen
            assert len(set(ncalls)) == 1, ncalls                              # the plan issues the ordinary path's all-reduces (4 SyncBN + the 2 gradient buckets)
            if mode == "1":
                plan = retrieval.step_plans_of(m)[0]
                names = plan.pl.op_names()
                for cb in ("allgather_targets", "allreduce_bn1", "allreduce_bn2", "clip_loss_data_parallel", "allreduce_bn2_bwd", "allreduce_early_bucket",
                           "allreduce_flat_gradient"):
                    assert cb in names, (cb, names)
                assert all(p.grad is None for p in m.parameters()) and float(m._engine().gflat.abs().max()) == 0.0
            out[mode] = ([float(a) for a in acc], {k: p.detach().clone().numpy() for k, p in m.named_parameters()}, int(correct))
    ret[rank] = out
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_plan_is_the_launch_by_launch_data_parallel_step():
    """VERDICT r5 #4a: the single plan under data parallelism -- cut into segments by host callbacks at the collectives (targets' all-gather, SyncBN exchanges,
    the loss between all-gather and reduce-scatter, the early gradient bucket, the final all-reduce).  Two gloo ranks, four steps (two ordinary, two through the
    plan) against the same four steps with EEGCLIP_STEP_PLAN=0: per-step losses, the final parameters of both ranks, and rank-identical parameters."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_plan_worker, args=(2, 29641, ret), nprocs=2, join=True)
    for r in range(2):
        (la, pa, ca), (lb, pb, cb) = ret[r]["1"], ret[r]["0"]
        np.testing.assert_allclose(la, lb, rtol=2e-5)
        assert ca == cb
        for k in pa:
            if k.endswith("key_projection.bias"):
                continue
            d = np.abs(pa[k] - pb[k])
            assert d.max() <= 4 * 3e-4 * 1.01 and (d > 5e-5).mean() <= 5e-3, (r, k, float(d.max()))
    for k in ret[0]["1"][1]:
        np.testing.assert_array_equal(ret[0]["1"][1][k], ret[1]["1"][1][k])      # the replicas stay identical


@pytest.mark.parametrize("mode", [(False, False), (False, True), (True, True)])
def test_clip_loss_gather_modes_match_reference_gloo_fixture(mode, golden):
    """B2: the three gather modes of models/loss.py:20-75, product ClipLoss under gloo vs the fixture recorded from the reference."""
    g = golden("dist_loss.npz")
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_loss_worker, args=(world, 29741 + 2 * int(mode[0]) + int(mode[1]), mode, ret), nprocs=world, join=True)
    tag = f"w{world}_ll{int(mode[0])}_gwg{int(mode[1])}"
    for r in range(world):
        assert abs(ret[r][0] - g[tag + "_loss"][r]) < 2e-5
        np.testing.assert_allclose(ret[r][1], g[tag + "_da"][r], atol=3e-6)
        np.testing.assert_allclose(ret[r][2], g[tag + "_db"][r], atol=6e-5)


@pytest.mark.parametrize("guided", [True, False])
def test_prior_sampling_chain_under_emulator_matches_oracle(guided):
    """Pipe.generate's restructured chain (time / condition embeddings hoisted out of the DDPM loop, skinny GEMM + fused stage tail per stage,
    classifier-free-guidance pair as one 2N-row pass) against the oracle's layer-by-layer chain with the same noise stream"""
    from oracle import prior as oprior
    state = syn.make_state(SEED + 20, oprior.prior_state_spec())
    P = oloops.torch_state(state)
    rng = np.random.default_rng(9)
    c = T(rng.standard_normal((2, 1024)).astype(np.float32)) if guided else None
    steps = 4
    with product_on_emulator():
        from eeg_image_decode_amd.prior import DiffusionPriorUNet, Pipe
        m = DiffusionPriorUNet(cond_dim=1024)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
        pipe = Pipe(m, device="cpu")
        h = pipe.generate(c_embeds=c, num_inference_steps=steps, guidance_scale=5.0, generator=torch.Generator().manual_seed(3))
    n = 2 if guided else 1
    hT = torch.randn(n, 1024, generator=torch.Generator().manual_seed(3))
    gen = torch.Generator().manual_seed(3)
    torch.randn(n, 1024, generator=gen)                     # the start latent comes first in the stream
    ho, _ = oprior.generate(P, oprior.DDPMSchedulerOracle(), c, steps, 5.0, generator=gen, h_T=hT)
    np.testing.assert_allclose(h.numpy(), ho.numpy(), atol=2e-4)


@pytest.mark.parametrize("cond", [True, False], ids=["conditioned", "unconditioned"])
def test_prior_training_step_on_plane_gemms_equals_the_fp32_operand_plans(cond, monkeypatch):
    """configs[3]'s training plans over bf16 planes (csrc/gemm_planes.hip, eegclip_wgrad_planes, fused stage tails; batch a multiple of 64) against the
    launch-per-Linear plans on fp32 operands (EEGCLIP_PRIOR_PLANES=0): same epsilon prediction, same gradient for every parameter, dropout masks
    included.  A narrow model so that the lane emulator finishes in seconds; the arithmetic (three split products, fp32 accumulate) is the same in both."""
    rng = np.random.default_rng(12)
    N, E, Cd = 64, 128, 64
    x, c = T(rng.standard_normal((N, E)).astype(np.float32)), T(rng.standard_normal((N, Cd)).astype(np.float32) * 4)
    tt = torch.from_numpy(rng.integers(0, 1000, N))
    w = T(rng.standard_normal((N, E)).astype(np.float32))
    res = {}
    with product_on_emulator():
        from eeg_image_decode_amd.prior import DiffusionPriorUNet
        for mode in ("1", "0"):
            monkeypatch.setenv("EEGCLIP_PRIOR_PLANES", mode)
            torch.manual_seed(5)
            m = DiffusionPriorUNet(embed_dim=E, cond_dim=Cd, hidden_dim=[128, 64, 64], time_embed_dim=64, dropout=0.2).train()
            torch.manual_seed(6)                                   # the forward draws its dropout seed from torch's generator
            out = m(x, tt, c if cond else None)
            (out * w).sum().backward()
            eng = m._engine()
            names = eng.plans[next(k for k in eng.plans if k[0] == "f")].op_names()
            assert ("eegclip_gemm_planes" in names) == (mode == "1") and ("eegclip_gemm_f32" in names) == (mode == "0")
            bn = eng.plans[next(k for k in eng.plans if k[0] == "b")].op_names()
            assert ("eegclip_wgrad_planes" in bn) == (mode == "1")
            res[mode] = (out.detach().clone(), {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()})
    o1, g1 = res["1"]
    o0, g0 = res["0"]
    np.testing.assert_allclose(o1.numpy(), o0.numpy(), atol=2e-5 * max(1.0, float(o0.abs().max())))
    for k in g0:
        if g0[k] is None:
            assert g1[k] is None, k
            continue
        assert g1[k] is not None, k
        np.testing.assert_allclose(g1[k].numpy(), g0[k].numpy(), atol=1e-6 + 2e-4 * float(g0[k].abs().max()), err_msg=k)


def test_prior_training_step_plan_is_the_launch_by_launch_training_loop(monkeypatch):
    """prior._TrainStepPlan (Pipe.train's steady-state iteration -- add_noise, forward, MSE, backward, clip_grad_norm_, Adam -- as ONE launch plan) against the
    same training run launch by launch (EEGCLIP_PRIOR_STEP_PLAN=0): 9 updates with whole-batch condition drops in between (both keys get a plan, the optimizer's
    launch sets are re-formed when the condition layers start to lag), the same generator calls in the same order, so the noise, the timesteps, the dropout
    seeds and the drop decisions are the same: per-update learning rates and drop decisions identical, the epoch losses and every parameter equal to
    round-off (the plan's optimizer launches clear the gradients behind their read; the arithmetic is launch for launch the loop's)."""
    rng = np.random.default_rng(21)
    N, E, Cd = 64, 128, 64
    data = [{"c_embedding": T(rng.standard_normal((N, Cd)).astype(np.float32)), "h_embedding": T(rng.standard_normal((N, E)).astype(np.float32))} for _ in range(9)]
    runs = {}
    with product_on_emulator():
        from eeg_image_decode_amd import prior as eprior
        built = []
        real_init = eprior._TrainStepPlan.__init__

        def counting_init(self, *a, **k):
            real_init(self, *a, **k)
            built.append(self.key)
        monkeypatch.setattr(eprior._TrainStepPlan, "__init__", counting_init)
        ran = []
        real_run = eprior._TrainStepPlan.run

        def counting_run(self, *a, **k):
            ran.append(self.key)
            return real_run(self, *a, **k)
        monkeypatch.setattr(eprior._TrainStepPlan, "run", counting_run)
        for mode in ("1", "0"):
            monkeypatch.setenv("EEGCLIP_PRIOR_STEP_PLAN", mode)
            torch.manual_seed(5)
            m = eprior.DiffusionPriorUNet(embed_dim=E, cond_dim=Cd, hidden_dim=[128, 64, 64], time_embed_dim=64, dropout=0.1)
            pipe = eprior.Pipe(m, device="cpu")
            pipe.cond_drop_prob = 0.3
            torch.manual_seed(11)
            losses = []
            import builtins
            monkeypatch.setattr(builtins, "print", lambda *a, **k: losses.append(a[0]))
            pipe.train(data, num_epochs=1, learning_rate=1e-3)
            runs[mode] = (list(pipe.lr_history), list(pipe.cond_dropped), losses[-1], {k: p.detach().clone() for k, p in m.named_parameters()})
            if mode == "1":
                assert any(pipe.cond_dropped) and not all(pipe.cond_dropped)          # the sequence holds both keys
                assert len(built) >= 2 and len({k[1] for k in built}) == 2, built     # ... and both went through a plan
                n_built, n_ran = len(built), len(ran)
                assert n_ran >= 3, (ran, pipe.cond_dropped)                            # ... that then carried the updates
        assert len(built) == n_built and len(ran) == n_ran                                                 # EEGCLIP_PRIOR_STEP_PLAN=0 builds none
    (lr1, d1, l1, p1), (lr0, d0, l0, p0) = runs["1"], runs["0"]
    assert lr1 == lr0 and d1 == d0
    assert abs(float(l1.split("loss:")[1]) - float(l0.split("loss:")[1])) < 1e-5
    for k in p0:
        np.testing.assert_allclose(p1[k].numpy(), p0[k].numpy(), atol=2e-6 + 2e-5 * float(p0[k].abs().max()), err_msg=k)


@pytest.fixture(scope="module")
def small_things_tree(tmp_path_factory):
    """the synthetic THINGS-EEG tree with the reference's class counts (they are hard-coded in its loader) but one channel pair and few samples"""
    root = str(tmp_path_factory.mktemp("things_small"))
    return root, syn.write_things_eeg_tree(root, 7, channels=2, n_times=58, feat_dim=8)


@pytest.mark.parametrize("train,kw,joint", [(True, dict(subjects=["sub-01", "sub-02"], exclude_subject="sub-02"), False),
                                            (False, dict(subjects=["sub-02"]), False),
                                            (True, dict(subjects=["sub-02", "sub-01"], time_window=[0.02, 0.05]), False),
                                            (True, dict(subjects=["sub-01", "sub-02"], exclude_subject="sub-02"), True),
                                            (False, dict(subjects=["sub-01", "sub-02"], exclude_subject="sub-02"), True)])
def test_dataset_staging_and_device_loader_under_emulator_match_oracle(train, kw, joint, small_things_tree):
    """datasets.EEGDataset (float64 file -> eegclip_stage_eeg -> resident float32 split) and its batch loader (eegclip_gather_rows) against the
    oracle's restatement of the reference class: whole data tensor, labels, texts, image paths, item tuples and shuffled batches"""
    from oracle import dataset as ods
    root, cfg = small_things_tree
    data, lab, texts, images, _, _ = ods.load_split(cfg["data_path"], cfg["img_directory_training" if train else "img_directory_test"], train=train,
                                                    joint=joint, **kw)
    with product_on_emulator():
        if joint:                                     # eegdatasets_joint_subjects.py: adap_subject keeps every training subject
            from eeg_image_decode_amd.datasets_joint import EEGDataset
            kw = dict(kw, adap_subject=kw["exclude_subject"])
            del kw["exclude_subject"]
        else:
            from eeg_image_decode_amd.datasets import EEGDataset
        ds = EEGDataset(cfg["data_path"], train=train, config=cfg, features_dir=root, device="cpu", **kw)
        assert len(ds) == len(data) and ds.text == texts and ds.img == images
        np.testing.assert_array_equal(ds.labels.numpy(), lab)
        if train:
            np.testing.assert_array_equal(ds.data.numpy(), data)                          # a cast and a copy: bit exact
        else:
            np.testing.assert_allclose(ds.data.numpy(), data, atol=1e-6)                  # mean over repetitions: float32 summation order
        for i in (0, 41, len(ds) - 1):
            x, label, text, tf, img, imf = ds[i]
            ti, ii = ods.item_rows(i, train, 1654 if train else 200)
            assert int(label) == lab[i] and text == texts[ti] and img == images[ii]
            assert torch.equal(tf, ds.text_features[ti]) and torch.equal(imf, ds.img_features[ii]) and torch.equal(x, ds.data[i])
        ld = ds.loader(batch_size=64, shuffle=True, drop_last=train, generator=torch.Generator().manual_seed(5))
        order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5)).numpy()
        assert len(ld) == (len(ds) // 64 if train else (len(ds) + 63) // 64)
        seen = 0
        for bi, (x, y, text, tf, img, imf) in enumerate(ld):
            idx = order[bi * 64:bi * 64 + 64]
            assert torch.equal(x, ds.data[idx]) and torch.equal(y, ds.labels[idx])
            rows = [ods.item_rows(int(i), train, 1654 if train else 200) for i in idx]
            assert text == [texts[r[0]] for r in rows] and img == [images[r[1]] for r in rows]
            assert torch.equal(tf, ds.text_features[[r[0] for r in rows]]) and torch.equal(imf, ds.img_features[[r[1] for r in rows]])
            seen += len(idx)
            if bi == 2:
                break
        assert seen == min(len(ds), 192)


def test_eval_mode_backward_uses_running_statistics():
    """model.eval() with grad enabled (frozen-BatchNorm fine-tuning, saliency).  BatchNorm normalises with the running statistics,
    so its backward has no batch-mean terms and the conv biases in front of it get real gradients -- against autograd through the oracle's
    eval-mode forward, input gradient included."""
    state_np = syn.make_state(SEED, oatms.state_spec())
    rng = np.random.default_rng(5)
    for k in list(state_np):                      # non-trivial running statistics (the synthetic state has mean 0 / var 1)
        if k.endswith("running_mean"):
            state_np[k] = (0.1 * rng.standard_normal(state_np[k].shape)).astype(np.float32)
        if k.endswith("running_var"):
            state_np[k] = (0.5 + rng.random(state_np[k].shape)).astype(np.float32)
    B = 3
    x0 = syn.eeg_batch(SEED + 43, B)
    img, txt = T(syn.unit_features(SEED + 43, B, tag="img")), T(syn.unit_features(SEED + 43, B, tag="txt"))
    with product_on_emulator():
        m = make_model(state_np).eval()
        x = T(x0).requires_grad_()
        z = m(x, 1)
        loss = 0.99 * m.loss_func(z, img, m.logit_scale) + 0.01 * m.loss_func(z, txt, m.logit_scale)
        loss.backward()
        grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}
        dx = x.grad.clone()
    tr = oloops.OracleTrainer(oloops.torch_state(state_np), p_scale=0.0)
    xo = T(x0).requires_grad_()
    zo = oatms.atms_forward(tr.P, xo, torch.full((B,), 1).long(), train=False)
    lo = oloss.mixed_loss(zo, img, txt, tr.P["logit_scale"])
    (dxo,) = torch.autograd.grad(lo, [xo])
    _, _, og, _ = tr.loss_and_grads(T(x0), torch.full((B,), 1).long(), img, txt, train=False)
    np.testing.assert_allclose(z.detach().numpy(), zo.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(dx.numpy(), dxo.numpy(), atol=3e-3 * float(dxo.abs().max()))
    for k, g in grads.items():
        if og[k] is None:
            assert g is None, k
        elif k != "encoder.encoder.attn_layers.0.attention.key_projection.bias":       # identically zero (softmax shift invariance)
            np.testing.assert_allclose(g.numpy(), og[k].numpy(), atol=1e-7 + 3e-3 * float(og[k].abs().max()), err_msg=k)
    for k in ("enc_eeg.0.tsconv.0.bias", "enc_eeg.0.tsconv.4.bias"):                    # not zero in eval mode
        assert float(og[k].abs().max()) > 1e-6 and float(grads[k].abs().max()) > 1e-6


def test_any_later_forward_at_the_same_batch_size_invalidates_a_pending_backward():
    """the activation buffers are shared by every plan of a batch size -- a no-grad / eval / other-token-branch forward between
    a training forward and its backward must make that backward raise, not produce silently wrong gradients"""
    from eeg_image_decode_amd._lib import EegclipError
    state_np = syn.make_state(SEED, oatms.state_spec())
    B = 2
    x = T(syn.eeg_batch(SEED + 44, B))
    img = T(syn.unit_features(SEED + 44, B, tag="img"))
    with product_on_emulator():
        m = make_model(state_np).train()
        for later in ("eval", "shared_token", "same"):
            m.train()
            z = m(x, 1)
            loss = m.loss_func(z, img, m.logit_scale)
            if later == "eval":
                with torch.no_grad():
                    m.eval()(x, 1)
            elif later == "shared_token":
                with torch.no_grad():
                    m(x, 10)
            else:
                with torch.no_grad():
                    m(x, 1)
            with pytest.raises(EegclipError):
                loss.backward()
        m.train()
        z = m(x, 1)                                   # a different batch size does not interfere
        with torch.no_grad():
            m(torch.cat([x, x]), 1)
        m.loss_func(z, img, m.logit_scale).backward()


def _prior_dp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import prior as oprior
    state = syn.make_state(SEED + 20, oprior.prior_state_spec())
    rng = np.random.default_rng(100 + rank)                       # every rank trains on its own shard
    data = [{"c_embedding": T(rng.standard_normal((4, 1024)).astype(np.float32)), "h_embedding": T(rng.standard_normal((4, 1024)).astype(np.float32))}
            for _ in range(3)]
    torch.manual_seed(1000 + rank)                                # ... with its own noise / timestep stream
    with product_on_emulator():
        from eeg_image_decode_amd.prior import DiffusionPriorUNet, Pipe
        m = DiffusionPriorUNet(cond_dim=1024)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
        pipe = Pipe(m, device="cpu")
        pipe.cond_drop_prob = 0.5
        pipe.train(data, num_epochs=2, learning_rate=1e-3)
        flat = m._engine().flat.clone()
    ret[rank] = (flat.numpy(), list(pipe.cond_dropped))
    dist.destroy_process_group()


def test_two_rank_prior_training_keeps_replicas_identical_through_condition_drops():
    """Pipe.train under data parallelism.  The whole-batch condition drop (diffusion_prior.py:304) is one shared decision: with
    per-rank draws a rank that dropped the condition would skip the Adam update of the condition layers while the others step them."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_prior_dp_worker, args=(world, 29761, ret), nprocs=world, join=True)
    (p0, d0), (p1, d1) = ret[0], ret[1]
    assert d0 == d1 and len(d0) == 6
    assert any(d0) and not all(d0), d0                               # both branches were exercised (p = 0.5, 6 updates, fixed seeds)
    np.testing.assert_array_equal(p0, p1)


def test_fused_adamw_state_dict_round_trip_and_reflatten():
    """optimizer.state_dict() carries exp_avg / exp_avg_sq per parameter (views of the flat moment buffers); a resumed optimizer and
    an optimizer whose model was re-flattened (model.float() re-creates the storage) continue exactly like the uninterrupted one"""
    state_np = syn.make_state(SEED, oatms.state_spec())
    B = 2
    xs = [T(syn.eeg_batch(SEED + 50 + i, B)) for i in range(3)]
    img = T(syn.unit_features(SEED + 50, B, tag="img"))

    def one(m, opt, x):
        opt.zero_grad()
        m.loss_func(m(x, 1), img, m.logit_scale).backward()
        opt.step()

    with product_on_emulator():
        from eeg_image_decode_amd import optim
        ref = make_model(state_np).train()
        opt_ref = optim.AdamW(ref.parameters(), lr=3e-4)
        for x in xs:
            one(ref, opt_ref, x)
        want = {k: p.detach().clone() for k, p in ref.named_parameters()}

        a = make_model(state_np).train()
        opt_a = optim.AdamW(a.parameters(), lr=3e-4)
        one(a, opt_a, xs[0])
        sd_model, sd_opt = {k: v.clone() for k, v in a.state_dict().items()}, opt_a.state_dict()
        st = sd_opt["state"]
        assert all("exp_avg" in v and "exp_avg_sq" in v and "step" in v for v in st.values()) and len(st) > 30
        b = make_model(state_np).train()                 # resume in a fresh model / optimizer
        b.load_state_dict(sd_model)
        opt_b = optim.AdamW(b.parameters(), lr=3e-4)
        opt_b.load_state_dict(sd_opt)
        one(b, opt_b, xs[1])
        b.float()                                        # nn.Module._apply: the engine re-flattens into a NEW storage on the next forward
        one(b, opt_b, xs[2])
        # (not bit-equal: split-K GEMMs add with atomics and Adam turns round-off-sized gradients into +-lr steps.  Run-to-run L1 distance of
        #  the whole parameter vector after these three steps: ~0.1; with the moments dropped at the resume: ~400)
        l1 = sum(float((p.detach() - want[k]).abs().sum()) for k, p in b.named_parameters())
        assert l1 < 4.0, l1
        assert len(opt_b._moments) == 1                  # the buffers of the abandoned storage are gone


@pytest.mark.parametrize("logits_dtype", ["f32", "bf16"])
def test_fused_clip_loss_matches_the_oracle_and_the_unfused_route(logits_dtype, monkeypatch):
    """ClipLoss on the fused kernels (n, D multiples of 64): value, d/dz, d/dtarget, d/dscale against the oracle's autograd (models/loss.py:
    122-140) -- parity mode within fp32 round-off class, throughput mode within its bf16 budget -- and the mixed two-target form"""
    rng = np.random.default_rng(11)
    n, Dm = 64, 128
    z0 = rng.standard_normal((n, Dm)).astype(np.float32)
    z0 = (z0 - z0.mean(1, keepdims=True)) / z0.std(1, keepdims=True)
    img0, txt0 = syn.unit_features(3, n, Dm, tag="fi"), syn.unit_features(3, n, Dm, tag="ft")
    tol = 3e-5 if logits_dtype == "f32" else 2e-2
    with product_on_emulator():
        from eeg_image_decode_amd import loss as ploss
        lf = ploss.ClipLoss(logits_dtype=logits_dtype)
        z, img, txt = T(z0).requires_grad_(), T(img0).requires_grad_(), T(txt0)
        sc = torch.tensor(2.6593, requires_grad=True)
        calls = []
        real = ploss.fused_infonce
        monkeypatch.setattr(ploss, "fused_infonce", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        loss = lf.forward_mixed(z, [(img, 0.99), (txt, 0.01)], sc)
        (3.0 * loss).backward()
        assert calls, "the fused route was not taken"
        got = [loss.detach().clone(), z.grad.clone(), img.grad.clone(), sc.grad.clone()]
    zo, io, to = T(z0).requires_grad_(), T(img0).requires_grad_(), T(txt0)
    so = torch.tensor(2.6593, requires_grad=True)
    lo = oloss.mixed_loss(zo, io, to, so)
    (3.0 * lo).backward()
    assert abs(float(got[0]) - float(lo)) < tol * max(1.0, abs(float(lo)))
    for g, r in ((got[1], zo.grad), (got[2], io.grad), (got[3], so.grad)):
        np.testing.assert_allclose(g.numpy(), r.numpy(), atol=(2e-3 if logits_dtype == "f32" else 5e-2) * float(r.abs().max()) + 1e-8)


def _fused_dp_worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, Dm = 64, 64
    a_all = T(syn.unit_features(SEED + 9, n * world, Dm, tag="da") * 8.0)
    b_all = T(syn.unit_features(SEED + 9, n * world, Dm, tag="db"))
    outs = []
    with product_on_emulator():
        from eeg_image_decode_amd.loss import ClipLoss
        for fused in ("1", "0"):
            os.environ["EEGCLIP_INFONCE_FUSED"] = fused
            a = a_all[rank * n:(rank + 1) * n].clone().requires_grad_(True)
            b = b_all[rank * n:(rank + 1) * n].clone().requires_grad_(True)
            sc = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
            l = ClipLoss(local_loss=mode[0], gather_with_grad=mode[1], rank=rank, world_size=world)(a, b, sc)
            l.backward()
            outs.append((float(l), a.grad.numpy().copy(), b.grad.numpy().copy(), float(sc.grad)))
    os.environ.pop("EEGCLIP_INFONCE_FUSED", None)
    ret[rank] = outs
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", [(False, False), (False, True), (True, True)])
def test_fused_clip_loss_gather_modes_equal_the_unfused_route_on_two_ranks(mode):
    """the three gather modes of models/loss.py:20-75 on the fused kernels (row-sharded blocks with the rank offset; full N x N on every rank)
    == the GEMM + log-sum-exp route, which the reference gloo fixture pins (test_clip_loss_gather_modes_match_reference_gloo_fixture)"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_fused_dp_worker, args=(world, 29801 + 2 * int(mode[0]) + int(mode[1]), mode, ret), nprocs=world, join=True)
    for r in range(world):
        (lf, daf, dbf, dsf), (lu, dau, dbu, dsu) = ret[r]
        assert abs(lf - lu) < 3e-5 * max(1.0, abs(lu))
        np.testing.assert_allclose(daf, dau, atol=2e-3 * np.abs(dau).max())
        np.testing.assert_allclose(dbf, dbu, atol=2e-3 * np.abs(dbu).max())
        assert abs(dsf - dsu) < 2e-3 * abs(dsu) + 1e-7


def test_sdxl_sampling_loop_under_emulator_matches_the_oracle():
    """F1 without a GPU: generate_ip_adapter_embeds on the smallest SDXL-shaped stand-in (8 x 8 latents, one layer per stage), 2 DDIM steps with
    classifier-free guidance -- gemm16, cross-attention, image projection, sampler step and the host loop together against oracle/sdxl_pipeline.py"""
    from sdxl_common import oracle_loop, small_pipe
    with product_on_emulator():
        pipe, W, cfg = small_pipe("ddim", "cpu", latent=8)
        emb = torch.randn(1, 1024, generator=torch.Generator().manual_seed(3)).half()
        out = pipe.generate_ip_adapter_embeds(prompt="", ip_adapter_embeds=emb, num_inference_steps=2, guidance_scale=5.0,
                                              generator=torch.Generator().manual_seed(11)).images
    ref = oracle_loop(pipe, W, cfg, "ddim", 2, 5.0, emb.float().numpy().astype(np.float64), seed=11)
    assert out.shape == (1, 4, 8, 8)
    d = np.abs(out.float().numpy() - ref)
    assert d.max() < 1e-2 * max(1.0, np.abs(ref).max()), (d.max(), np.abs(ref).max())


def test_processor_kv_cache_never_serves_another_tensors_projections_under_emulator():
    from sdxl_common import check_processor_kv_cache_never_serves_another_tensors_projections
    with product_on_emulator():
        check_processor_kv_cache_never_serves_another_tensors_projections("cpu")


def test_sdxl_schedulers_reproduce_their_defining_identities():
    """host-side coefficients of the two schedulers (product) against the oracle's step on random data, plus the DDIM identity: a latent built from
    (x0, eps) at timestep t steps to the same (x0, eps) mixture at the previous timestep"""
    from eeg_image_decode_amd.sdxl import DDIMScheduler, EulerAncestralDiscreteScheduler
    from oracle import sdxl_pipeline as osp
    rng = np.random.default_rng(0)
    x, eps, nz = rng.standard_normal(64), rng.standard_normal(64), rng.standard_normal(64)
    d, od = DDIMScheduler(), osp.DDIM(50)
    d.set_timesteps(50)
    assert d.timesteps.tolist() == od.timesteps.tolist() == list(range(981, 0, -20))
    for i, t in enumerate(d.timesteps.tolist()):
        cx, ce, cn = d.coefficients(t)
        od.i = i
        np.testing.assert_allclose(cx * x + ce * eps, od.step(eps, x), rtol=1e-12, atol=1e-12)
        a_t = od.acp[t]
        a_p = od.acp[t - 20] if t - 20 >= 0 else od.acp[0]
        x0 = rng.standard_normal(64)
        xt = np.sqrt(a_t) * x0 + np.sqrt(1 - a_t) * eps
        np.testing.assert_allclose(cx * xt + ce * eps, np.sqrt(a_p) * x0 + np.sqrt(1 - a_p) * eps, atol=1e-12)
    for n in (1, 4):
        e, oe = EulerAncestralDiscreteScheduler(), osp.EulerAncestral(n)
        e.set_timesteps(n)
        assert e.timesteps.tolist() == oe.timesteps.tolist() and e.timesteps[0] == 999
        assert abs(e.init_noise_sigma - oe.init_noise_sigma) < 1e-12
        for i in range(n):
            cx, ce, cn = e.coefficients(i)
            oe.i = i
            np.testing.assert_allclose(cx * x + ce * eps + cn * nz, oe.step(eps, x, nz), rtol=1e-10, atol=1e-10)
        assert e.coefficients(n - 1)[2] == 0.0                    # the last step lands on sigma = 0: no noise is added


@pytest.mark.parametrize("variant", ["retrieval", "reconstruction", "joint_uniform", "joint_mixed"])
def test_single_submission_step_plan_is_the_launch_by_launch_step(monkeypatch, variant):
    """step_plan.StepPlan (the whole steady-state step -- forward, accuracy readout, fused InfoNCE, backward, AdamW -- replayed by ONE eegclip_plan_run call)
    against retrieval's launch-by-launch step FROM THE SAME STATE: after two ordinary steps (they create the plans and the optimizer's launch cache) the
    model / optimizer / RNG state is snapshotted, the third step runs through the plan, then again the ordinary way from the snapshot.  Same features, loss
    and accuracy count (computed before the update); the parameters agree as two runs of the ordinary path do.  (With a single-threaded emulator, where float
    atomics are ordered, four such steps are bit-identical: tools/check_step_plan_bitwise.py.)  B = 32: the plan takes batches that are not whole 64-tiles through the small InfoNCE form.
    Variants: the reconstruction objective (Generation/ATMS_reconstruction.py:222-228: one InfoNCE target + an MSE term), the joint-subject
    model (Retrieval/ATMS_retrieval_joint_train.py:172-192) on one-subject batches and on a batch that mixes four subjects."""
    import copy
    joint = variant.startswith("joint")
    objective = "reconstruction" if variant == "reconstruction" else "retrieval"
    alpha = 0.9 if objective == "reconstruction" else 0.99
    B, NC = 32, 40
    cls = T(syn.unit_features(SEED + 4, NC, tag="c"))
    rng = np.random.default_rng(1)
    data = [(T(syn.eeg_batch(SEED + 100 + i, B)), T(syn.unit_features(SEED + 200 + i, B, tag="i")), T(syn.unit_features(SEED + 300 + i, B, tag="t")),
             T(rng.integers(0, NC, size=B).astype(np.int64))) for i in range(3)]
    sid = {"joint_uniform": 3, "joint_mixed": rng.integers(1, 5, B).tolist()}.get(variant, 1)

    def new_model():
        if joint:
            from eeg_image_decode_amd.retrieval_joint import ATMS as JointATMS
            torch.manual_seed(7)
            m = JointATMS(joint_train=True)
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            return m.train()
        return make_model(syn.make_state(SEED, oatms.state_spec())).train()

    with product_on_emulator():
        from eeg_image_decode_amd import optim, retrieval, step_plan
        monkeypatch.setattr(step_plan, "_runtime_ok", lambda: True)
        monkeypatch.setattr(step_plan, "_on_device", lambda t: True)
        monkeypatch.setattr(step_plan.StepPlan, "WARM_STEPS", 2)
        if variant == "reconstruction":
            monkeypatch.setenv("EEGCLIP_WGRAD_ADAMW", "1")
        torch.manual_seed(5)
        m = new_model()
        opt = optim.AdamW(m.parameters(), lr=3e-4)
        acc, correct = [], torch.zeros(1, dtype=torch.int32)
        for x, img, txt, lab in data[:2]:
            retrieval.contrastive_step(m, opt, x, sid, img, txt, lab, cls, acc, correct, alpha=alpha, objective=objective)
        assert not retrieval.step_plans_of(m)           # warm-up: the ordinary path
        snap = (copy.deepcopy(m.state_dict()), copy.deepcopy(opt.state_dict()), torch.get_rng_state(), correct.clone())
        x, img, txt, lab = data[2]
        f_plan = retrieval.contrastive_step(m, opt, x, sid, img, txt, lab, cls, acc, correct, alpha=alpha, objective=objective)
        plans = retrieval.step_plans_of(m)
        assert len(plans) == 1 and isinstance(plans[0], step_plan.StepPlan)             # the third step went through the plan ...
        names = plans[0].pl.op_names()
        assert names.count("eegclip_adamw_step_zero_grad") >= 1 and ("eegclip_infonce_small_grad" in names or "eegclip_infonce_fused_fwd" in names)
        # the step ends in the optimizer; opt-in (here: the reconstruction variant) the slab reduction of the block's weight gradients steps it itself
        assert names[-1] == ("eegclip_wgrad_tok_reduce_adamw" if variant == "reconstruction" else "eegclip_adamw_step_zero_grad"), names[-3:]
        assert ("eegclip_mse_loss_grad_scaled" in names) == (objective == "reconstruction")
        assert all(p.grad is None for p in m.parameters()) and float(m._engine().gflat.abs().max()) == 0.0
        res_plan = ({k: p.detach().clone() for k, p in m.named_parameters()}, float(acc[-1]), int(correct), f_plan.clone(),
                    {k: v.clone() for k, v in m.state_dict().items() if "running" in k})
        # ... and once more the ordinary way, from the same state
        monkeypatch.setenv("EEGCLIP_STEP_PLAN", "0")
        m2 = new_model()
        m2.load_state_dict(snap[0])
        opt2 = optim.AdamW(m2.parameters(), lr=3e-4)
        opt2.load_state_dict(snap[1])
        torch.set_rng_state(snap[2])
        acc2, correct2 = [], snap[3].clone()
        f_ord = retrieval.contrastive_step(m2, opt2, x, sid, img, txt, lab, cls, acc2, correct2, alpha=alpha, objective=objective)
        assert not retrieval.step_plans_of(m2)
    np.testing.assert_allclose(res_plan[3].numpy(), f_ord.numpy(), atol=2e-5)            # (the head's split-K atomics: unordered under the multi-threaded emulator)
    assert abs(res_plan[1] - float(acc2[-1])) < 2e-5 * max(1.0, abs(res_plan[1])) and abs(res_plan[2] - int(correct2)) <= 1
    for k, v in res_plan[4].items():
        np.testing.assert_allclose(v.numpy(), m2.state_dict()[k].numpy(), rtol=1e-5, atol=1e-6)
    for k, p in m2.named_parameters():
        if k.endswith("key_projection.bias"):
            continue
        d = np.abs(res_plan[0][k].numpy() - p.detach().numpy())
        assert d.max() <= 3e-4 * 1.01 and (d > 2e-5).mean() <= 2e-3, (k, float(d.max()), float((d > 2e-5).mean()))


def test_step_plan_after_a_keep_grads_step_does_not_accumulate_onto_stale_gradients(monkeypatch):
    """contrastive_step(keep_grads=True) -- documented for gradient-norm logging -- leaves the gradients in the flat buffer; the next call's
    zero_grad(set_to_none=True) drops the .grad views but not the values.  A plan step accumulates into that buffer without attach_grads(): it must clear it
    first.  Sequence: 2 warm steps, 1 plan step, 1 keep_grads step (ordinary path), 1 plan step -- against the same five steps with EEGCLIP_STEP_PLAN=0."""
    state_np = syn.make_state(SEED, oatms.state_spec())
    B, NC = 32, 40
    cls = T(syn.unit_features(SEED + 4, NC, tag="c"))
    rng = np.random.default_rng(2)
    data = [(T(syn.eeg_batch(SEED + 400 + i, B)), T(syn.unit_features(SEED + 500 + i, B, tag="i")), T(syn.unit_features(SEED + 600 + i, B, tag="t")),
             T(rng.integers(0, NC, size=B).astype(np.int64))) for i in range(5)]
    keep = (False, False, False, True, False)
    runs = []
    with product_on_emulator():
        from eeg_image_decode_amd import optim, retrieval, step_plan
        monkeypatch.setattr(step_plan, "_runtime_ok", lambda: True)
        monkeypatch.setattr(step_plan, "_on_device", lambda t: True)
        monkeypatch.setattr(step_plan.StepPlan, "WARM_STEPS", 2)
        for mode in ("1", "0"):
            monkeypatch.setenv("EEGCLIP_STEP_PLAN", mode)
            torch.manual_seed(5)
            m = make_model(state_np).train()
            opt = optim.AdamW(m.parameters(), lr=3e-4)
            acc, correct = [], torch.zeros(1, dtype=torch.int32)
            seen = []
            for (x, img, txt, lab), kg in zip(data, keep):
                retrieval.contrastive_step(m, opt, x, 1, img, txt, lab, cls, acc, correct, keep_grads=kg)
                if kg:
                    assert float(m._engine().gflat.abs().max()) > 0.0 and m.proj_eeg[0].weight.grad is not None      # the gradients were kept
                seen.append(bool(retrieval.step_plans_of(m)))
            if mode == "1":
                assert seen == [False, False, True, True, True]
                assert float(m._engine().gflat.abs().max()) == 0.0              # the last plan step ended with the fused clear
            runs.append(([float(a) for a in acc], {k: p.detach().clone() for k, p in m.named_parameters()}))
    np.testing.assert_allclose(runs[0][0], runs[1][0], rtol=2e-5)
    for k, p in runs[1][1].items():
        if k.endswith("key_projection.bias"):
            continue
        d = np.abs(runs[0][1][k].numpy() - p.numpy())
        assert d.max() <= 5 * 3e-4 * 1.01 and (d > 5e-5).mean() <= 5e-3, (k, float(d.max()), float((d > 5e-5).mean()))      # stale gradients would move everything by ~lr


def test_joint_subject_model_with_more_subjects_than_one_weight_gradient_launch_holds():
    """eegclip_wgrad_tok takes <= 12 problems per launch; a joint-subject model with a larger subject table (atms.ATMS(table_subjects=14):
    the reference takes any num_subjects, Embed.py:127-131) must still train -- it takes the grouped-GEMM plans instead of the fused block.  Checked against
    a 10-subject model carrying the same value embeddings for the subjects of the batch: same embeddings, same gradients."""
    ids14, ids10 = [13, 0, 13, 5], [3, 0, 3, 5]
    B = 4
    x0 = T(syn.eeg_batch(SEED + 43, B))
    img, txt = T(syn.unit_features(SEED + 43, B, tag="img")), T(syn.unit_features(SEED + 43, B, tag="txt"))
    with product_on_emulator():
        from eeg_image_decode_amd import atms
        torch.manual_seed(3)
        big = atms.ATMS(joint_train=True, table_subjects=14)
        small = atms.ATMS(joint_train=True, table_subjects=10)
        sd_b, sd_s = big.state_dict(), small.state_dict()
        for k in sd_s:
            if ".value_embedding." in k:
                s_ = int(k.split(".value_embedding.")[1].split(".")[0])
                sd_s[k] = sd_b[k.replace(f".value_embedding.{s_}.", f".value_embedding.{13 if s_ == 3 else s_}.")].clone()
            elif sd_s[k].shape == sd_b[k].shape:
                sd_s[k] = sd_b[k].clone()
            else:                                              # the subject-token table: rows of the subjects present
                sd_s[k] = sd_b[k][:sd_s[k].shape[0]].clone()
                sd_s[k][3] = sd_b[k][13]
        small.load_state_dict(sd_s)
        res = []
        for m, ids in ((big, ids14), (small, ids10)):
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            m.train()
            z = m(x0, torch.tensor(ids))
            (0.99 * m.loss_func(z, img, m.logit_scale) + 0.01 * m.loss_func(z, txt, m.logit_scale)).backward()
            names = m._engine().plans[next(k for k in m._engine().plans if k[0] == "f")].op_names()
            res.append((z.detach().clone(), {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}, names))
    assert "eegclip_token_block_fwd" not in res[0][2] and "eegclip_gemm_f32_grouped" in res[0][2]          # 14 subjects: launch-per-Linear plans
    assert "eegclip_token_block_fwd" in res[1][2]
    np.testing.assert_allclose(res[0][0].numpy(), res[1][0].numpy(), atol=1e-4)
    gb, gs = res[0][1], res[1][1]
    for s_big, s_small in ((13, 3), (0, 0), (5, 5)):
        for leaf in ("weight", "bias"):
            a, b = gb[f"encoder.enc_embedding.value_embedding.{s_big}.{leaf}"], gs[f"encoder.enc_embedding.value_embedding.{s_small}.{leaf}"]
            np.testing.assert_allclose(a.numpy(), b.numpy(), atol=1e-7 + 3e-3 * float(b.abs().max()))
    assert gb["encoder.enc_embedding.value_embedding.7.weight"] is None and gb["encoder.enc_embedding.value_embedding.12.weight"] is None
    np.testing.assert_allclose(gb["proj_eeg.0.weight"].numpy(), gs["proj_eeg.0.weight"].numpy(), atol=3e-3 * float(gs["proj_eeg.0.weight"].abs().max()))
