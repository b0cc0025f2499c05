"""Contrastive train / eval loops with the reference's signatures (Retrieval/ATMS_retrieval.py:199-512), driven by
the HIP encoder + loss.  Differences that do NOT change results: running loss / accuracy are accumulated on the device
and read back once per epoch (the reference calls .item() twice per step: two host syncs), and the k-way evaluation
scores every query with one GEMM + a top-k kernel instead of a Python loop of 1xk matmuls.  ``random.sample`` is consumed
in exactly the reference's order, so a seeded run reproduces the reference's candidate sets.
"""
import ctypes
import os
import random
import re
import warnings
import weakref

import torch

from . import _abi, step_plan as _step_plan_module  # noqa: F401  (loaded with this module: the test emulator patches every loaded module)
from ._lib import check, cuda_available, current_stream, lib, raw_stream, require_cuda, use_stream

D = _abi.dim


def extract_id_from_string(s):
    m = re.search(r'\d+$', s)
    return int(m.group()) if m else None


def _stream():
    return raw_stream()


def scaled_logits(z, feats, precision=_abi.PREC_F32):
    """raw z @ feats.T (the monotone logit_scale factor is irrelevant for argmax / top-k and is applied by the caller where values matter).
    precision: exact fp32 products by default (evaluation: the ranking the reference's fp32 matmul produces); the running TRAIN accuracy of the
    batch loop passes the step's GEMM arithmetic (split-bf16: logits within ~3e-5, half the time of a launch that shares the GPU with the backward)."""
    z, feats = z.contiguous(), feats.contiguous()
    n, dm = z.shape
    c = feats.shape[0]
    out = torch.empty(n, c, dtype=torch.float32, device=z.device)
    d = _abi.GemmDesc(M=n, N=c, K=dm, A=z.data_ptr(), Am=D(dm), Ak=D(1), B=feats.data_ptr(), Bk=D(1), Bn=D(dm), C=out.data_ptr(),
                      Cm=D(c), Cn=D(1), Cpre=None, bias_n=None, bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=0, act=0,
                      drop_p=0.0, seed=0, drop_site=0, split_k=1, precision=precision)
    check(lib().eegclip_gemm_f32(ctypes.byref(d), _stream()), "logits gemm")
    return out


def topk_rows(logits, k, scale=None):
    """scale: optional device scalar; ranking is by scale*x (sign handled on the device, no host sync)"""
    n, c = logits.shape
    out = torch.empty(n, k, dtype=torch.long, device=logits.device)
    sp = scale.detach().reshape(1) if torch.is_tensor(scale) else None
    check(lib().eegclip_topk_rows(logits.data_ptr(), n, c, logits.stride(0), k, sp.data_ptr() if sp is not None else None,
                                  out.data_ptr(), _stream()), "topk")
    return out


def topk_retrieval(z, class_feats, logit_scale, k, precision=_abi.PREC_F32):
    """indices of the k best classes per query, ties -> lowest index (ATMS_retrieval.py:246 argmax, :320 topk)."""
    require_cuda(z, "z")
    logits = scaled_logits(z.detach().float(), class_feats.detach().float(), precision)
    if not torch.is_tensor(logit_scale):
        logit_scale = torch.full((1,), float(logit_scale), dtype=torch.float32, device=z.device)
    return topk_rows(logits, k, logit_scale)


def _uniform_ids(batch_size, subject_id, device):
    if not isinstance(subject_id, int):           # a per-sample id list (batches that mix subjects: the joint-subject model's general case)
        import numpy as np
        host = np.asarray(subject_id.cpu() if isinstance(subject_id, torch.Tensor) else subject_id, dtype=np.int64)      # (one C loop: a list comprehension +
        ids = torch.from_numpy(host).to(device, non_blocking=True)                                                       #  torch.tensor(list) were 50 us per step)
        ids._eegclip_host_ids = host              # ATMS.forward lays the batch out by subject from the host copy: no device->host sync
        return ids
    key = (batch_size, subject_id, str(device))
    ids = _UNIFORM_IDS.get(key)
    if ids is None:                               # (one fill per (batch size, subject), not one per step; nobody writes to it)
        if len(_UNIFORM_IDS) > 64:
            _UNIFORM_IDS.clear()
        ids = torch.full((batch_size,), subject_id, dtype=torch.long, device=device)
        ids._eegclip_uniform_id = subject_id      # lets ATMS.forward pick the token branch without a device->host sync
        _UNIFORM_IDS[key] = ids
    return ids


_UNIFORM_IDS = {}
_GC_SETTLED = False


def settle_gc(force=False):
    """One full collection, then gc.freeze(): everything alive now -- the module graphs of torch and the libraries it pulled in, millions of tracked
    objects -- moves to the permanent generation and is never walked again.  Python's generation-2 collection otherwise walks all of it at an
    unpredictable step of an epoch: 75 ms on the MI355X host (tools/host_stalls.py), as long as 80 training steps, with the GPU queue running dry
    behind it.  Called once per process by the training loops (train_model, Pipe.train) before their first step; semantics are unchanged (objects
    created afterwards are collected as usual)."""
    global _GC_SETTLED
    if _GC_SETTLED and not force:
        return
    import gc
    gc.collect()
    gc.freeze()
    _GC_SETTLED = True


_SIDE = {}


def _side_stream(device):
    """second HIP stream per device for work nobody waits for inside the step (the running train-accuracy readout)"""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device, priority=0)
    return _SIDE[key]


def contrastive_step(eeg_model, optimizer, eeg_data, subject_id, img_features, text_features, labels, class_feats, loss_acc, correct,
                     alpha=0.99, objective="retrieval", keep_grads=False):
    """One iteration of the reference batch loop (ATMS_retrieval.py:209-250) with every tensor already on the device:
    forward, image + text InfoNCE (0.99/0.01), backward, optimizer step, running loss and train-accuracy -- no host sync.
    With an optimizer that offers it (optim.AdamW / Adam: supports_step_and_zero_grad) the update also performs the zero_grad() that opens the
    next iteration, so the gradients are None when this returns -- the reference loop leaves them in place until its next zero_grad()
    (ATMS_retrieval.py:209-231): pass keep_grads=True to get exactly that (gradient-norm logging, clipping diagnostics) at the price of a separate
    clearing pass; `loss_acc` is a device scalar (added to in place) or a list (appended to).
    Under torch.distributed (world > 1) the loss gathers embeddings across ranks and the flat gradient is averaged."""
    from . import dist as edist
    if not _GC_SETTLED:
        settle_gc()
    optimizer.zero_grad()
    overlap = edist.world_size() > 1 and hasattr(eeg_model, "_engine") and os.environ.get("EEGCLIP_DP_OVERLAP", "1") != "0"
    prev_overlap = getattr(eeg_model, "overlap_grad_allreduce", False)
    if overlap:
        eeg_model.overlap_grad_allreduce = True          # one backward per step HERE: its early gradient bucket may be reduced while it still runs
    try:
        return _contrastive_step(eeg_model, optimizer, eeg_data, subject_id, img_features, text_features, labels, class_feats, loss_acc, correct,
                                 alpha, objective, keep_grads)
    finally:
        if overlap:
            eeg_model.overlap_grad_allreduce = prev_overlap      # (a caller accumulating gradients over several backwards must not inherit it)


_STEP_PLANS_MAX = 16
# model -> {configuration: state}.  Keyed weakly by the model and holding neither the model nor the optimizer strongly (StepPlan keeps weak references
# too): a dropped model frees its plans, buffers and engine -- nothing here closes a reference cycle through the engine, which retrieval.settle_gc()'s
# gc.freeze() would make immortal for the first model of a process
_STEP_PLAN_TABLES = weakref.WeakKeyDictionary()


def step_plans_of(model):
    """the StepPlan objects built for `model` so far (tests, bench.py)"""
    return [st["plan"] for st in _STEP_PLAN_TABLES.get(model, {}).values() if st["plan"]]


def _step_plan(eeg_model, optimizer, eeg_data, subject_id, img_features, text_features, labels, class_feats, alpha, objective, keep_grads):
    """(state, StepPlan or None): the single-submission plan of this steady-state step (step_plan.py) once the ordinary path has run it
    StepPlan.WARM_STEPS times -- those steps create the encoder plans, the activation buffers and the optimizer's launch cache the plan is built
    from.  state is the per-configuration counter the caller advances after an ordinary step."""
    from . import dist as edist
    from .step_plan import NotApplicable, StepPlan
    if not StepPlan.eligible(eeg_model, optimizer, eeg_data, subject_id, img_features, text_features, labels, class_feats, objective, keep_grads,
                             edist.world_size()):
        return None, None
    eng = eeg_model._engine()
    table = _STEP_PLAN_TABLES.get(eeg_model)
    if table is None:
        table = _STEP_PLAN_TABLES[eeg_model] = {}
    if isinstance(subject_id, int):
        subj_key = subject_id
    else:       # joint-subject model with per-sample ids: the optimizer's launch set depends on WHICH subjects are present (their value embeddings are live)
        import numpy as np
        subj_key = tuple(np.unique(np.asarray(subject_id.cpu() if isinstance(subject_id, torch.Tensor) else subject_id, dtype=np.int64)).tolist())
    key = (id(optimizer), eeg_data.shape[0], float(alpha), class_feats.shape[0], subj_key, objective, edist.world_size())
    st = table.get(key)
    if st is not None and st["opt"]() is not optimizer:               # (an id reused by another optimizer)
        st = None
    if st is None:
        if len(table) >= _STEP_PLANS_MAX:
            table.clear()
        st = table[key] = {"warm": 0, "plan": None, "opt": weakref.ref(optimizer)}
    sp = st["plan"]
    if sp is not None and sp is not False:
        if sp.still_valid(eeg_model, optimizer) and (eeg_model.joint_train or subject_id >= 10 or eng.bufs[eeg_data.shape[0]].get("ids_uniform") == subject_id):
            return st, sp
        st["plan"], st["warm"] = None, 0             # something changed under the plan: warm up again on the ordinary path
        return st, None
    if sp is None and st["warm"] >= StepPlan.WARM_STEPS and optimizer._fast_last.get(0) is not None and all(p.grad is None for p in optimizer.param_groups[0]["params"]):
        try:
            st["plan"] = StepPlan(eeg_model, optimizer, eeg_data.shape[0], alpha, class_feats.shape[0], objective, edist.world_size())
            return st, st["plan"]
        except NotApplicable as e:                    # (anything else is a bug in the plan builder and propagates)
            st["plan"] = False                        # this configuration does not have the pieces (e.g. launch-per-Linear plans): ordinary path for good
            warnings.warn(f"single-submission step plan not applicable, keeping the launch-by-launch path: {e}", RuntimeWarning, stacklevel=3)
    return st, None


def _contrastive_step(eeg_model, optimizer, eeg_data, subject_id, img_features, text_features, labels, class_feats, loss_acc, correct, alpha, objective,
                      keep_grads=False):
    from . import dist as edist
    st, sp = _step_plan(eeg_model, optimizer, eeg_data, subject_id, img_features, text_features, labels, class_feats, alpha, objective, keep_grads)
    if sp is not None:
        feats, loss = sp.run(eeg_data, img_features, text_features, labels, class_feats, correct, subject_id)
        if isinstance(loss_acc, list):
            loss_acc.append(loss)
        else:
            loss_acc += loss
        return feats
    if st is not None:
        st["warm"] += 1
    batch_size = eeg_data.size(0)
    subject_ids = _uniform_ids(batch_size, subject_id, eeg_data.device)
    loss_func = eeg_model.loss_func
    if edist.world_size() > 1 and hasattr(loss_func, "gather_targets"):
        # the targets are inputs: their all-gather ([img | txt] in ONE collective) runs under the encoder forward instead of after it
        loss_func.gather_targets(*((img_features,) if objective == "reconstruction" else (img_features, text_features)))
    eeg_features = eeg_model(eeg_data, subject_ids).float()
    logit_scale = eeg_model.logit_scale
    # running train accuracy (ATMS_retrieval.py:241-250: logits against all class features, argmax, count): it only needs the
    # forward output, so it runs on a second stream underneath the loss / backward / optimizer kernels instead of after them
    side = _side_stream(eeg_data.device) if cuda_available() else None
    main = current_stream() if side is not None else None
    if side is not None:
        side.wait_stream(main)
        with use_stream(side):
            _accumulate_accuracy(eeg_features, class_feats, logit_scale, labels, batch_size, correct)
    if objective == "reconstruction":
        # Generation/ATMS_reconstruction.py:222-228 (alpha = 0.9 there): 10 * (alpha * MSE(z, img) + (1 - alpha) * ClipLoss(z, img)); the text
        # loss the reference also evaluates never enters the objective
        from .loss import mse_loss
        loss = mse_loss(eeg_features, img_features, 10.0 * alpha) + 10.0 * (1 - alpha) * loss_func(eeg_features, img_features, logit_scale)
    elif hasattr(loss_func, "forward_mixed"):        # both targets in one pass: one gradient w.r.t. the EEG features, one accumulator
        loss_func._unit_upstream_grad = True         # `loss` IS the scalar backward() is called on three lines below
        try:
            loss = loss_func.forward_mixed(eeg_features, [(img_features, alpha), (text_features, 1 - alpha)], logit_scale)
        finally:
            loss_func._unit_upstream_grad = False
    else:                                            # a user-supplied loss module: the reference's two calls (ATMS_retrieval.py:224-229)
        loss = alpha * loss_func(eeg_features, img_features, logit_scale) + (1 - alpha) * loss_func(eeg_features, text_features, logit_scale)
    loss.backward()
    if edist.world_size() > 1:
        edist.average_flat_grads(eeg_model.flat_parameters()[1], eeg_model._engine() if hasattr(eeg_model, "_engine") else None)
    if side is not None:
        main.wait_stream(side)             # join BEFORE the optimizer rewrites logit_scale, which the readout's ranking kernel reads
    else:
        _accumulate_accuracy(eeg_features, class_feats, logit_scale, labels, batch_size, correct)
    if getattr(optimizer, "supports_step_and_zero_grad", False) and not keep_grads:
        optimizer.step(zero_grad=True)     # the update and the zero_grad() that opens the next iteration in one pass over the gradients
    else:
        optimizer.step()
    if isinstance(loss_acc, list):
        loss_acc.append(loss.detach())     # summed when the epoch ends (running_loss): no launch per step
    else:
        loss_acc += loss.detach()
    return eeg_features.detach()


def running_loss(loss_acc):
    """the epoch's summed loss (ATMS_retrieval.py:232 `total_loss += loss.item()`) from what contrastive_step accumulated: a device scalar
    (legacy: added to in place, one launch per step) or a list of per-step loss scalars (one stack + sum here)"""
    if isinstance(loss_acc, list):
        return torch.stack([l.reshape(()) for l in loss_acc]).sum() if loss_acc else torch.zeros(())
    return loss_acc


def _accumulate_accuracy(eeg_features, class_feats, logit_scale, labels, batch_size, correct):
    from .plan import default_gemm_precision
    logits = scaled_logits(eeg_features.detach().float(), class_feats.detach().float(), default_gemm_precision())
    sc = logit_scale.detach().reshape(1) if torch.is_tensor(logit_scale) else torch.full((1,), float(logit_scale), dtype=torch.float32, device=logits.device)
    check(lib().eegclip_top1_count(logits.data_ptr(), batch_size, logits.shape[1], logits.stride(0), sc.data_ptr(), labels.data_ptr(), correct.data_ptr(),
                                   _stream()), "top1_count")


def train_model(sub, eeg_model, dataloader, optimizer, device, text_features_all, img_features_all, config, objective="retrieval", alpha=0.99):
    eeg_model.train()
    settle_gc()
    text_features_all = text_features_all.to(device).float()
    img_features_all = (img_features_all[::10]).to(device).float().contiguous()
    features_list = []
    loss_acc = []                                          # per-step loss scalars, summed once at the end (running_loss)
    correct = torch.zeros(1, dtype=torch.int32, device=device)
    total, n_batches = 0, 0
    subject_id = extract_id_from_string(sub)
    for batch_idx, (eeg_data, labels, text, text_features, img, img_features) in enumerate(dataloader):
        eeg_data = eeg_data.to(device, non_blocking=True).float()
        text_features = text_features.to(device, non_blocking=True).float()
        img_features = img_features.to(device, non_blocking=True).float()
        labels = labels.to(device, non_blocking=True).long()
        features_list.append(contrastive_step(eeg_model, optimizer, eeg_data, subject_id, img_features, text_features, labels,
                                              img_features_all, loss_acc, correct, alpha=alpha, objective=objective))
        total += eeg_data.size(0)
        n_batches += 1
    average_loss = float(running_loss(loss_acc)) / n_batches            # the only host syncs of the epoch
    accuracy = int(correct) / total
    return average_loss, accuracy, torch.cat(features_list, dim=0)


def evaluate_model(sub, eeg_model, dataloader, device, text_features_all, img_features_all, k, config):
    eeg_model.eval()
    text_features_all = text_features_all.to(device).float()
    img_features_all = img_features_all.to(device).float().contiguous()
    alpha = 0.99
    all_labels = set(range(text_features_all.size(0)))
    subject_id = extract_id_from_string(sub)
    loss_acc = torch.zeros((), dtype=torch.float32, device=device)
    # evaluation is per process (the reference evaluates in one process): a loss module configured for data-parallel TRAINING would all-gather
    # here -- a hang if only rank 0 evaluates, duplicated positives scored as negatives if every rank evaluates the same test set
    loss_func = eeg_model.loss_func
    if getattr(loss_func, "world_size", 1) > 1:
        from .loss import ClipLoss
        loss_func = ClipLoss(logits_dtype=getattr(loss_func, "logits_dtype", "f32"))
    feats, labs, cands = [], [], []
    n_batches = 0
    with torch.no_grad():
        for batch_idx, (eeg_data, labels, text, text_features, img, img_features) in enumerate(dataloader):
            labels_host = labels.tolist() if labels.device.type == "cpu" else labels.cpu().tolist()
            eeg_data = eeg_data.to(device).float()
            text_features = text_features.to(device).float()
            img_features = img_features.to(device).float()
            batch_size = eeg_data.size(0)
            eeg_features = eeg_model(eeg_data, _uniform_ids(batch_size, subject_id, device))
            logit_scale = eeg_model.logit_scale
            img_loss = loss_func(eeg_features, img_features, logit_scale)
            text_loss = loss_func(eeg_features, text_features, logit_scale)
            loss_acc += img_loss * alpha + text_loss * (1 - alpha)
            n_batches += 1
            for label in labels_host:
                if k not in (200, 100, 50, 10, 4, 2):
                    print("Error.")
                    continue
                possible_classes = list(all_labels - {label})
                selected_classes = random.sample(possible_classes, k - 1) + [label]
                cands.append(selected_classes)                      # these index the features that are scored ...
                if k != 200:
                    random.sample(possible_classes, k - 1)          # ... the reference re-samples AFTER gathering (:328); label stays last
                labs.append(label)
            feats.append(eeg_features)
        z = torch.cat(feats, 0)
        n = z.shape[0]
        full = scaled_logits(z, img_features_all)                                   # (n, n_classes) raw dot products
        sel = torch.tensor(cands, dtype=torch.long, device=device)                  # (n, k) candidate class ids
        cand_logits = torch.gather(full, 1, sel).contiguous()
        kk = min(5, k)
        top = topk_rows(cand_logits, kk, eeg_model.logit_scale).cpu()                                      # positions within each candidate list
    total = n
    correct = int((top[:, 0] == k - 1).sum())                                       # the label is always the last candidate
    top5_correct = int((top == k - 1).any(dim=1).sum()) if k in (200, 100, 50) else 0
    average_loss = float(loss_acc) / max(1, n_batches)
    return average_loss, correct / total, top5_correct / total


def get_eegfeatures(sub, eeg_model, dataloader, device):
    """Feature dump used to condition the diffusion prior (Generation notebooks, cell 2-3): eval-mode embeddings."""
    eeg_model.eval()
    subject_id = extract_id_from_string(sub)
    out = []
    with torch.no_grad():
        for batch in dataloader:
            x = batch[0].to(device).float()
            out.append(eeg_model(x, _uniform_ids(x.size(0), subject_id, device)))
    return torch.cat(out, 0)


def main_train_loop(sub, current_time, eeg_model, train_dataloader, test_dataloader, optimizer, device, text_features_train_all,
                    text_features_test_all, img_features_train_all, img_features_test_all, config, logger=None):
    """Epoch loop + 6 k-way evaluations per epoch; returns the reference's list of per-epoch dicts (:410-424).
    `logger` is optional here (the reference dereferences it unconditionally)."""
    results = []
    best_accuracy = 0.0
    for epoch in range(config.epochs):
        train_loss, train_accuracy, _ = train_model(sub, eeg_model, train_dataloader, optimizer, device, text_features_train_all,
                                                    img_features_train_all, config=config)
        if (epoch + 1) % 5 == 0 and getattr(config, "save_checkpoints", True):
            base = f"./models/contrast/{config.encoder_type}/{sub}/{current_time}" if getattr(config, "insubject", True) \
                else f"./models/contrast/across/{config.encoder_type}/{current_time}"
            os.makedirs(base, exist_ok=True)
            torch.save(eeg_model.state_dict(), f"{base}/{epoch + 1}.pth")
        ev = lambda kk: evaluate_model(sub, eeg_model, test_dataloader, device, text_features_test_all, img_features_test_all, k=kk, config=config)
        test_loss, test_accuracy, top5_acc = ev(200)
        _, v2_acc, _ = ev(2)
        _, v4_acc, _ = ev(4)
        _, v10_acc, _ = ev(10)
        _, v50_acc, v50_top5_acc = ev(50)
        _, v100_acc, v100_top5_acc = ev(100)
        results.append({"epoch": epoch + 1, "test_loss": test_loss, "test_accuracy": test_accuracy, "v2_acc": v2_acc, "v4_acc": v4_acc,
                        "v10_acc": v10_acc, "top5_acc": top5_acc, "v50_acc": v50_acc, "v100_acc": v100_acc,
                        "v50_top5_acc": v50_top5_acc, "v100_top5_acc": v100_top5_acc})
        best_accuracy = max(best_accuracy, test_accuracy)
        if logger is not None:
            logger.log({"Train Loss": train_loss, "Train Accuracy": train_accuracy, "Test Loss": test_loss, "Test Accuracy": test_accuracy,
                        "v2 Accuracy": v2_acc, "v4 Accuracy": v4_acc, "v10 Accuracy": v10_acc, "Epoch": epoch})
        print(f"Epoch {epoch + 1}/{config.epochs} - Train Loss: {train_loss:.4f}, Train Accuracy: {train_accuracy:.4f}, "
              f"Test Loss: {test_loss:.4f}, Test Accuracy: {test_accuracy:.4f}, Top5 Accuracy: {top5_acc:.4f}")
    return results
