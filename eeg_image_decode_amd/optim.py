"""Fused AdamW / Adam for the flat-buffer models (drop-in for ``torch.optim.AdamW(model.parameters(), lr)``,
Retrieval/ATMS_retrieval.py:548, and ``optim.Adam`` of Generation/diffusion_prior.py:286).

Parameters of ATMS / DiffusionPriorUNet are views into one flat fp32 buffer and so are their gradients.  step() groups
the parameters that received a gradient into maximal contiguous runs and issues ONE eegclip_adamw_step launch per run
(normally 2 per step instead of ~45 tiny foreach kernels).  Semantics are torch's: decoupled weight decay, bias
correction with a per-parameter step count, parameters whose grad is None are skipped entirely (no decay, no step).

The moments live in two flat buffers shaped like the parameter storage; ``state[p]["exp_avg"]`` / ``["exp_avg_sq"]`` are VIEWS of
them, so ``state_dict()`` / ``load_state_dict()`` carry step, exp_avg and exp_avg_sq per parameter exactly like torch.optim.AdamW
(loaded tensors are copied into the flat buffers at the next step; so are the old moments after the model was re-flattened).
"""
import torch

from ._lib import EegclipError, check, lib, raw_stream, require_cuda


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._runs_cache = {}
        self._fast = {}              # param group index -> {gradient-tensor identities -> that set's launches} (see step())
        self._fast_last = {}         # param group index -> the set the last step used (its lazy step counts are pending)
        self._moments = {}           # storage base ptr -> (m, v) shaped like the whole flat storage
        self.grad_scale_dev = None   # optional device scalar multiplied into every gradient (clipping)

    def _moments_for(self, p):
        st = p.untyped_storage()
        key = st.data_ptr()
        if key not in self._moments:
            n = st.nbytes() // 4
            self._moments[key] = (torch.zeros(n, dtype=torch.float32, device=p.device), torch.zeros(n, dtype=torch.float32, device=p.device))
        return self._moments[key]

    def _link_state(self, live):
        """make state[p]["exp_avg"/"exp_avg_sq"] views of the flat moment buffers; moments that live elsewhere (loaded by load_state_dict, or
        views of the buffers of a storage the model has since left) are copied in first.  Runs only when the set of live tensors changes."""
        for p in live:
            m, v = self._moments_for(p)
            off = (p.data_ptr() - p.untyped_storage().data_ptr()) // 4
            st = self.state[p]
            for name, buf in (("exp_avg", m), ("exp_avg_sq", v)):
                view = buf[off:off + p.numel()].view(p.shape)
                old = st.get(name)
                if old is not None and old.data_ptr() != view.data_ptr():
                    view.copy_(old.to(device=view.device, dtype=torch.float32).reshape(p.shape))
                st[name] = view
        alive = {p.untyped_storage().data_ptr() for g in self.param_groups for p in g["params"]}
        for key in [k for k in self._moments if k not in alive]:
            del self._moments[key]                        # buffers of storages no parameter lives in any more

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._runs_cache.clear()                          # the loaded moments are linked (copied into the flat buffers) at the next step
        self._fast.clear()
        self._fast_last.clear()

    supports_step_and_zero_grad = True

    def zero_grad(self, set_to_none=True):
        """torch's zero_grad with the common case first: after step(zero_grad=True) every .grad is already None and the batch loop's opening
        optimizer.zero_grad() (ATMS_retrieval.py:209) has nothing to do -- one pass over the parameters instead of torch's per-parameter
        bookkeeping (~20 us of host time per step)"""
        if set_to_none and all(p.grad is None for g in self.param_groups for p in g["params"]):
            return
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        """zero_grad=True: optimizer.step() and the optimizer.zero_grad() that opens the next iteration (ATMS_retrieval.py:209,231) as ONE pass
        over the gradients -- the update kernel clears each gradient behind its read, then .grad is set to None exactly as
        zero_grad(set_to_none=True) would; owners of a flat gradient buffer (the ATMS engine) are told their buffer is already clear."""
        loss = closure() if closure is not None else None
        L = lib()
        fn = L.eegclip_adamw_step_zero_grad if zero_grad else L.eegclip_adamw_step
        stream = raw_stream()
        gs = self.grad_scale_dev.data_ptr() if self.grad_scale_dev is not None else None
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            grads = [p.grad for p in params]
            b1, b2 = group["betas"]
            # steady state of a training loop: the very same gradient tensors as at an earlier step (views of a flat gradient buffer; the cache
            # holds them, so an equal id is the same object): every check below already passed and the launches are unchanged -- ~100 us of
            # per-parameter bookkeeping per step otherwise.  Per-parameter step counts are brought up to date lazily (_flush_steps).  Several
            # such sets are remembered (the joint-subject model: the value embeddings that are live change with the subjects of the batch, and
            # the reference's joint loop alternates subjects batch by batch); switching between them settles the lazy counts first.
            table = self._fast.setdefault(gi, {})
            fast = table.get(tuple(map(id, grads)))
            if fast is not None and all(a is b for a, b in zip(grads, fast["grads"])) and \
                    all(p0.grad.data_ptr() == gp and p0.data_ptr() == wp for (p0, n, wp, gp, mp, vp, members) in fast["launch"]):
                if self._fast_last.get(gi) is not fast:
                    self._flush_steps(gi)
                    # the runs were formed from parameters with EQUAL step counts; another set may since have stepped only some of them
                    steps_now = [self.state[p0]["step"] for (p0, *_rest) in fast["launch"]]
                    if all(self.state[q]["step"] == st for (_p0, _n, _w, _g, _m, _v, members), st in zip(fast["launch"], steps_now) for q in members):
                        fast["run_steps"] = steps_now
                        self._fast_last[gi] = fast
                    else:
                        table.pop(tuple(map(id, grads)), None)
                        fast = None
            else:
                fast = None
            if fast is not None:
                fast["run_steps"] = [st + 1 for st in fast["run_steps"]]
                fast["pending"] += 1
                for (p0, n, wp, gp, mp, vp, members), st in zip(fast["launch"], fast["run_steps"]):
                    check(fn(wp, gp, mp, vp, n, group["lr"], b1, b2, group["eps"], group["weight_decay"], st, 1.0, gs, stream), "adamw_step")
                if zero_grad:
                    for own, ptrs in fast["owners"]:
                        own.grads_cleared(ptrs)
                continue
            self._flush_steps(gi)
            self._fast_last[gi] = None
            live = [p for p in params if p.grad is not None]
            if not live:
                continue
            for p in live:
                require_cuda(p, "parameter")
                if p.dtype != torch.float32:
                    raise EegclipError("eeg_image_decode_amd.optim works on float32 parameters")
                st = self.state[p]
                st["step"] = st.get("step", 0) + 1
            ck = (id(group), tuple((id(p), p.data_ptr(), p.grad.data_ptr()) for p in live))
            cached = self._runs_cache.get(ck)
            if cached is not None and all(self.state[q]["step"] == self.state[p0]["step"] for (p0, n, members) in cached for q in members):
                runs = [(p0, n, self.state[p0]["step"], members) for (p0, n, members) in cached]
            else:
                self._link_state(live)
                runs = self._make_runs(live)
                if len(self._runs_cache) >= 64:           # (joint-subject training: the live set changes with the subjects of the batch)
                    self._runs_cache.clear()
                self._runs_cache[ck] = [(p0, n, members) for (p0, n, _, members) in runs]
            launch = []
            for (p0, n, step, members) in runs:
                m, v = self._moments_for(p0)
                off = (p0.data_ptr() - p0.untyped_storage().data_ptr()) // 4
                launch.append((p0, n, p0.data_ptr(), p0.grad.data_ptr(), m.data_ptr() + 4 * off, v.data_ptr() + 4 * off, members))
                check(fn(*launch[-1][2:6], n, group["lr"], b1, b2, group["eps"], group["weight_decay"], step, 1.0, gs, stream), "adamw_step")
            owners = {}
            for p in live:
                own = getattr(p, "_eegclip_grad_owner", None)
                own = own() if own is not None else None
                if own is not None:
                    owners.setdefault(id(own), (own, []))[1].append(p.grad.data_ptr())
            owners = [(own, frozenset(ptrs)) for own, ptrs in owners.values()]
            if zero_grad:
                for own, ptrs in owners:
                    own.grads_cleared(ptrs)
            if len(table) >= 64:
                table.clear()
            fast = dict(grads=grads, live=live, launch=launch, owners=owners, run_steps=[r[2] for r in runs], pending=0)
            table[tuple(map(id, grads))] = fast
            self._fast_last[gi] = fast
        if zero_grad:
            for group in self.param_groups:                  # zero_grad(set_to_none=True) for every parameter, stepped or not
                for p in group["params"]:
                    p.grad = None
        return loss

    def activate_launch_set(self, gi, fast):
        """make `fast` (one of this optimizer's cached launch sets) the set whose lazy step counts are pending -- what step() does when the live set changes
        (the joint-subject model alternates subjects batch by batch) -- for a caller that replays the set's launches itself (step_plan.StepPlan).  False:
        the set is gone, or its runs no longer share step counts (the caller takes the ordinary path, which rebuilds it)."""
        if self._fast_last.get(gi) is fast:
            return True
        if not any(f is fast for f in self._fast.get(gi, {}).values()):
            return False
        self._flush_steps(gi)
        steps_now = [self.state[p0]["step"] for (p0, *_rest) in fast["launch"]]
        if not all(self.state[q]["step"] == st for (_p0, _n, _w, _g, _m, _v, members), st in zip(fast["launch"], steps_now) for q in members):
            return False
        fast["run_steps"] = steps_now
        self._fast_last[gi] = fast
        return True

    def _flush_steps(self, gi=None):
        """bring state[p]["step"] up to date with the steps taken on the fast path"""
        for g in ([gi] if gi is not None else list(self._fast_last)):
            fast = self._fast_last.get(g)
            if fast and fast["pending"]:
                for p in fast["live"]:
                    self.state[p]["step"] += fast["pending"]
                fast["pending"] = 0

    def state_dict(self):
        self._flush_steps()
        return super().state_dict()

    def __getstate__(self):
        """copy.deepcopy / pickle go through here, not through state_dict(): state[p]["step"] is LAZY on the fast path (only the per-group counter
        advances; the per-parameter counts are brought up to date by _flush_steps), so flush first -- a copy restored from stale counts would apply
        the wrong bias correction.  torch's Optimizer.__getstate__ keeps defaults / state / param_groups only."""
        self._flush_steps()
        return super().__getstate__()

    def __setstate__(self, state):
        """the copy starts without launch caches and flat moment buffers (raw device addresses of the ORIGINAL's tensors): its moments are the
        per-parameter tensors of `state`, linked into fresh flat buffers at its first step (as after load_state_dict)"""
        super().__setstate__(state)
        self._runs_cache, self._fast, self._fast_last, self._moments = {}, {}, {}, {}
        if not hasattr(self, "grad_scale_dev"):
            self.grad_scale_dev = None

    def _make_runs(self, live):
        """maximal runs of parameters that are contiguous (up to 12 bytes of alignment padding) in BOTH the weight and
        the gradient storage and share a step count -> [(first param, numel incl. padding, step, member parameters)]"""
        items = sorted(live, key=lambda p: p.data_ptr())
        runs = []
        cur = None
        for p in items:
            step = self.state[p]["step"]
            if cur is not None:
                p0, end_w, end_g, st0, members = cur
                gap_w, gap_g = p.data_ptr() - end_w, p.grad.data_ptr() - end_g
                same = p.untyped_storage().data_ptr() == p0.untyped_storage().data_ptr()
                if same and st0 == step and 0 <= gap_w <= 12 and gap_w == gap_g:
                    members.append(p)
                    cur = (p0, p.data_ptr() + 4 * p.numel(), p.grad.data_ptr() + 4 * p.numel(), st0, members)
                    continue
                runs.append((p0, (end_w - p0.data_ptr()) // 4, st0, members))
            cur = (p, p.data_ptr() + 4 * p.numel(), p.grad.data_ptr() + 4 * p.numel(), step, [p])
        p0, end_w, _, st0, members = cur
        runs.append((p0, (end_w - p0.data_ptr()) // 4, st0, members))
        return runs


class Adam(AdamW):
    """torch.optim.Adam semantics (no weight decay) on the same fused kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0.0)
