"""One submission per training step (round 5; widened in round 6).

The reference's batch loop (Retrieval/ATMS_retrieval.py:209-250) is paced by the Python interpreter: forward, two losses, backward, optimizer, the
running accuracy -- every torch op issued from the loop.  In steady state nothing about the step changes except a handful of pointers (the batch, its
targets and labels, the fresh output tensor), the dropout seed and the optimizer's step count, so the WHOLE step is one plan:

    [encoder forward; riders of its 1x1-conv launch split this step's head weights and loss targets into planes]
    -> logits of all targets (one K-parallel plane GEMM) -> InfoNCE rows / columns -> gradient matrices as planes -> query-gradient GEMM (slabs)
       [fork: accuracy GEMM from planes + top-1 count]
    -> [encoder backward; the optimizer update of the early gradient bucket on the second stream behind the conv stack's last gradient]
    -> join -> fused AdamW + gradient clear over the rest

replayed by ONE foreign call (eegclip_plan_run).  The launches, their order, their arguments and the random-number consumption are those of the
launch-by-launch path -- tools/check_step_plan_bitwise.py (single-threaded emulator: ordered float atomics) compares every tensor bit for bit,
tests/test_product_on_emulator.py one step from a common snapshot at round-off tolerance for the four objectives, tests/test_full_size_gpu.py the plan's
losses, embeddings and post-AdamW parameters against the ORACLE at B = 256, tests/test_dp_gpu.py the data-parallel form against the launch-by-launch
data-parallel loop -- so this is a host-side optimisation only.  Objectives: retrieval, reconstruction (Generation/ATMS_reconstruction.py:222-228),
the joint-subject model (Retrieval/ATMS_retrieval_joint_train.py:172-192; the per-step subject layout is patched into the plan), and -- world > 1 -- data
parallelism (the op array is cut into segments around host callbacks: target all-gather, the data-parallel loss, the flat-gradient all-reduce).
Anything outside the steady state (first steps of a run, another batch size, a user-supplied loss or optimizer, keep_grads) takes the ordinary path;
EEGCLIP_STEP_PLAN=0 disables the plan, NotApplicable says why one cannot be built.
"""
import ctypes
import os
import weakref

import torch

from . import _abi
from ._lib import cuda_available, lib, raw_stream
from .plan import Plan

D = _abi.dim


def enabled():
    return os.environ.get("EEGCLIP_STEP_PLAN", "1") != "0"


def _runtime_ok():
    return cuda_available()


def _on_device(t):
    return t.is_cuda


class NotApplicable(Exception):
    """this configuration does not have the pieces a step plan is built from (launch-per-Linear encoder plans, no optimizer launch cache ...): the caller
    keeps the ordinary path.  Anything else raised while a plan is built is a bug and propagates."""


class StepPlan:
    """the steady-state contrastive step of one (model, optimizer, batch size, loss mix) as a single launch plan"""

    WARM_STEPS = 3          # ordinary steps before the plan is built: they create the encoder plans, the activation buffers and the optimizer's launch cache

    @property
    def eng(self):
        return self._eng_ref()

    def __init__(self, model, optimizer, B, alpha, n_classes, objective="retrieval", world=1):
        from . import loss as eloss
        from .atms import P_DIM
        eng = model._engine()
        self.world = int(world)
        # the loss mix: retrieval = alpha * ClipLoss(z, img) + (1 - alpha) * ClipLoss(z, txt)  (ATMS_retrieval.py:224-234); reconstruction =
        # 10 * (alpha * MSE(z, img) + (1 - alpha) * ClipLoss(z, img))  (Generation/ATMS_reconstruction.py:222-228): one InfoNCE target + an MSE term whose
        # gradient joins the query gradient as one more slab of the backward's upstream gradient
        self.objective = objective
        self.weights = (float(alpha), 1.0 - float(alpha)) if objective == "retrieval" else (10.0 * (1.0 - float(alpha)),)
        self.mse_w = 0.0 if objective == "retrieval" else 10.0 * float(alpha)
        T_ = len(self.weights)
        self.T = T_
        self.joint = bool(eng.joint)
        # (weak: the plan hangs off the engine, which hangs off the model -- a strong reference back would be a cycle that retrieval.settle_gc()'s
        #  gc.freeze() makes immortal for the first model of a process)
        self._model_ref, self._opt_ref = weakref.ref(model), weakref.ref(optimizer)
        self._eng_ref = weakref.ref(eng)
        self.B, self.alpha, self.n_classes = B, float(alpha), n_classes
        key = eng.last_key
        self.key = key
        Bk, train, shared, probs, W = key
        if not (Bk == B and train and W == (self.world if model.sync_batchnorm else 1)):
            raise NotApplicable("the last forward was not a training forward of this job at this batch size")
        early = self.world > 1 and os.environ.get("EEGCLIP_DP_OVERLAP", "1") != "0"
        self.bwd_key = ("b", B, train, shared, probs, False, W, early)
        self.fwd = eng.plans.get(("f",) + key)
        self.bwd = eng.plans.get(self.bwd_key)
        if self.fwd is None or self.bwd is None or self.fwd.tb_desc is None or self.bwd.x_gemm is not None or optimizer._fast_last.get(0) is None:
            raise NotApplicable("the fused transformer-block plans and the optimizer's launch cache are required")
        if self.world > 1 and (self.joint or objective != "retrieval"):
            raise NotApplicable("data-parallel plans cover the retrieval objective of the single-subject model")
        if self.joint and not hasattr(self.bwd, "j_wk_ops"):
            raise NotApplicable("joint-subject model: the per-subject weight gradients must be the fused block's (eegclip_wgrad_tok) launches")
        self.probs = probs
        dev = eng.device
        self.dev = dev
        L = lib()
        Dm = P_DIM
        self.planes = 2 if model.loss_func.logits_dtype == "f32" else 1
        pl = Plan(f"contrastive_step[B={B}]", precision=self.fwd.precision)
        pl._keep += [self.fwd, self.bwd]

        def splice(src, lo=0, hi=None, at=None):
            """ops [lo, hi) of `src` appended; op i of src lands at (at if given else the current end) + i - lo ... i.e. `at` = plan index of src op 0 + shift"""
            hi = len(src.ops) if hi is None else hi
            base = len(pl.ops) - lo if at is None else at
            assert base + lo == len(pl.ops)
            for fn, args, name, side in src.ops[lo:hi]:
                pl.ops.append((fn, list(args), name, side))
            pl._seed_slots += [(base + i, j) for i, j in src._seed_slots if lo <= i < hi]
            if lo == 0:
                pl._seed_descs += src._seed_descs
            return base

        # ---- the targets' operand planes: inputs only.  With the head on plane GEMMs (round 6) their split RIDES in the forward's 1x1-conv launch next to the
        # head weights' (csrc/split_rider.h: no launch, no second stream, no join); otherwise a second-stream launch under the encoder's forward (items below)
        self.on_planes = bool(self.world == 1 and getattr(self.fwd, "head_planes", False) and getattr(self.bwd, "head_planes", False) and
                              eloss.head_gemm_enabled(B, T_ * B, Dm))
        if self.world > 1:
            # ---- data parallel (round 6): the same single plan, cut into segments by HOST CALLBACKS at the collectives -- the targets' all-gather (started
            # before the forward), the SyncBN exchanges inside the spliced encoder plans, the loss (all-gather of the embeddings, the rank's row-sharded
            # InfoNCE blocks, reduce-scatter of the gathered-copy gradients: loss._ClipLossFn's own code, called without autograd), the early gradient
            # bucket's asynchronous all-reduce and the final one.  Every segment between two callbacks is one eegclip_plan_run call.
            self._build_data_parallel(model, optimizer, eng, pl, splice, B, n_classes, Dm, L, dev)
            return
        if self.mse_w and not self.on_planes:
            raise NotApplicable("the reconstruction objective's plan needs the head on plane GEMMs (the MSE gradient is a slab of the upstream gradient)")
        self.split_op = None
        if not self.on_planes:
            self.split_op = len(pl.ops)
            pl.ops.append((L.eegclip_split_rows, [None, T_, None], "eegclip_split_rows", os.environ.get("EEGCLIP_START_SIDE", "1") != "0"))
        # ---- encoder forward
        f0 = splice(self.fwd)
        self.fwd_base = f0
        self.out_op = f0 + self.fwd.out_op
        # ---- running train accuracy: raw z @ class_feats^T, top-1, count: second stream.  The ops are emitted INSIDE the backward, right behind its first
        # second-stream launch (the head LayerNorm's parameter gradients), so that they share that fork instead of paying one of their own
        sc_ptr = model.logit_scale.detach().reshape(1).data_ptr()
        ncp = (n_classes + 3) // 4 * 4
        self.acc_on_planes = bool(getattr(self.fwd, "head_planes", False)) and os.environ.get("EEGCLIP_HEAD_GEMM", "1") != "0"
        if self.acc_on_planes:
            # (round 6) from planes on csrc/head_gemm.hip: the query features leave the head's LayerNorm as planes anyway, the class table is split ONCE per
            # table (run()); the fp32-operand GEMM took 30 us alone and 90 us beside the conv stack's backward
            self.logits = torch.empty(B, ncp, dtype=torch.float32, device=dev)
            self.class_planes = torch.zeros(2, ncp, Dm, dtype=torch.bfloat16, device=dev)       # (rows >= n_classes: zero padding up to a multiple of 4 columns)
            self.acc_desc = None
        else:
            self.logits = torch.empty(B, n_classes, dtype=torch.float32, device=dev)
            self.acc_desc = _abi.GemmDesc(M=B, N=n_classes, K=Dm, A=0, Am=D(Dm), Ak=D(1), B=0, Bk=D(1), Bn=D(Dm), C=self.logits.data_ptr(), Cm=D(n_classes), Cn=D(1),
                                          Cpre=None, bias_n=None, bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=0, act=0, drop_p=0.0, seed=0,
                                          drop_site=0, split_k=1, precision=self.fwd.precision)
            pl._keep.append(self.acc_desc)

        def accuracy_ops():
            if self.acc_on_planes:
                pl.call_desc("eegclip_head_gemm", _abi.HeadGemmDesc(a_hi=self.q_planes[0].data_ptr(), a_lo=self.q_planes[1].data_ptr(), b_hi=self.class_planes[0].data_ptr(),
                                                                   b_lo=self.class_planes[1].data_ptr(), lda=Dm, ldb=Dm, M=B, N=ncp, K=Dm, slices=1, slab_stride=0,
                                                                   C=self.logits.data_ptr(), ldc=ncp), side=True)
            else:
                pl.ops.append((L.eegclip_gemm_f32, [ctypes.byref(self.acc_desc), None], "eegclip_gemm_f32", True))
            self.count_op = len(pl.ops)
            pl.call("eegclip_top1_count", self.logits.data_ptr(), B, n_classes, self.logits.shape[1], sc_ptr, 0, 0, side=True)

        self.q_planes = torch.empty(2, B, Dm, dtype=torch.bfloat16, device=dev)                 # the query features, written by the head's LayerNorm
        # ---- image + text InfoNCE on the fused kernels (loss.py: _ClipLossFn.forward, the W == 1 fused branch).  Operand planes: the targets are inputs of the
        # step -- split (and, for the query gradient, split TRANSPOSED: round 6) by second-stream launches at the very start; the query features leave the
        # head's LayerNorm as planes (the same rounding as eegclip_split_rows), so no split launch sits between the forward and the loss
        self.t_planes = torch.empty(2, T_ * B, Dm, dtype=torch.bfloat16, device=dev)            # hi | lo of [img; txt]
        self.stack = None if self.on_planes else torch.empty(T_ * B, Dm, dtype=torch.float32, device=dev)     # round 5's fp32 right-hand operand of the query gradient (loss.py)
        if self.on_planes:
            self.items = (_abi.SplitItem * 4)()                    # this plan's own rider table: the forward plan's weights item + the targets
            self.items[0] = self.fwd.rider_items[0]
            self.item0 = 1
            pl._keep.append(self.items)
            pl.set_arg(self.fwd_base + self.fwd.rider_op, 23, self.items)
            pl.set_arg(self.fwd_base + self.fwd.rider_op, 24, 1 + T_)
        else:
            self.items = (_abi.SplitItem * 2)()
            self.item0 = 0
            pl._keep.append(self.items)
            pl.ops[self.split_op][1][0] = self.items
        for i in range(T_):
            self.items[self.item0 + i] = _abi.SplitItem(src=0, hi=self.t_planes[0, i * B:].data_ptr(), lo=self.t_planes[1, i * B:].data_ptr(), rows=B, cols=Dm,
                                                        ld_src=Dm, ld_out=Dm, transpose=0, copy=self.stack[i * B].data_ptr() if self.stack is not None else None,
                                                        ld_copy=Dm)
        fn, args, name, side = pl.ops[self.out_op]
        if name == "eegclip_residual_layernorm_fwd_slabs":              # (round 6: the head's LayerNorm adds the second Linear's slabs; planes are arguments 19 / 20)
            args[19], args[20] = self.q_planes[0].data_ptr(), self.q_planes[1].data_ptr()
        else:
            if name != "eegclip_residual_layernorm_fwd":
                raise NotApplicable(f"unexpected output op {name}")
            pl.ops[self.out_op] = (L.eegclip_residual_layernorm_fwd_planes, args[:-1] + [self.q_planes[0].data_ptr(), self.q_planes[1].data_ptr(), None],
                                   "eegclip_residual_layernorm_fwd_planes", side)
        ws = int(L.eegclip_infonce_fused_workspace_floats(B, B))
        self.if_buf = torch.empty(2 * T_ * (ws + 2 * B), dtype=torch.float32, device=dev)
        if self.on_planes:
            self.G = None
            self.Gp = torch.empty(2, B, T_ * B, dtype=torch.bfloat16, device=dev)           # [G_img | G_txt] side by side, as hi | lo planes
        else:
            self.G = torch.empty(B, T_ * B, dtype=torch.float32, device=dev)
        base = self.if_buf.data_ptr()

        def planes_of(i):
            hi, lo = (self.q_planes[0], self.q_planes[1]) if i == 0 else (self.t_planes[0, (i - 1) * B:], self.t_planes[1, (i - 1) * B:])
            return (hi.data_ptr(), lo.data_ptr() if self.planes == 2 else None)

        ap = planes_of(0)
        arr = (_abi.InfonceProblem * (2 * T_))()
        for t_, w in enumerate(self.weights):
            bp = planes_of(1 + t_)
            for j, (q, k) in enumerate(((ap, bp), (bp, ap))):
                o = base + 4 * (2 * t_ + j) * (ws + 2 * B)
                arr[2 * t_ + j] = _abi.InfonceProblem(q_hi=q[0], q_lo=q[1], k_hi=k[0], k_lo=k[1], col0=0, weight=0.5 * w, part=o, diag=o + 4 * ws,
                                                      lse=o + 4 * (ws + B), lse_k=None, G=None, ldg=0)
        garr = (_abi.InfonceProblem * T_)()
        for t_ in range(T_):
            garr[t_] = arr[2 * t_]
            if self.on_planes:
                garr[t_].G, garr[t_].ldg = None, T_ * B
                garr[t_].G_hi, garr[t_].G_lo = self.Gp[0, :, t_ * B:].data_ptr(), self.Gp[1, :, t_ * B:].data_ptr()
            else:
                garr[t_].G, garr[t_].ldg = self.G[:, t_ * B:].data_ptr(), T_ * B
            garr[t_].lse_k = arr[2 * t_ + 1].lse
            garr[t_].part_k, garr[t_].diag_k = arr[2 * t_ + 1].part, arr[2 * t_ + 1].diag
        pl._keep += [arr, garr]
        # the loss reads planes (and, round 5's form, the fp32 stack) that the SECOND stream wrote at the start of the step: ordered by the forward plan's join
        # in front of the conv stack when it has one -- an explicit join otherwise.  With riders nothing ran on the second stream yet.
        if self.split_op is not None and "join" not in [op[2] for op in pl.ops[self.fwd_base:]]:
            pl.join()
        # (round 6) the accuracy readout forks HERE, right behind the forward: its ~30 us of second-stream work overlap the loss and the head's backward -- small
        # launches that leave most of the chip idle -- instead of queueing behind the conv stack's backward, whose workgroups own the CUs' LDS (the plane
        # GEMM took 83 us there).  EEGCLIP_ACC_EARLY=0: inside the backward's first fork (round 5's place; A/B aid)
        self.acc_early = os.environ.get("EEGCLIP_ACC_EARLY", "1") != "0" and self.acc_on_planes
        if self.acc_early:
            accuracy_ops()
        self.small = self.on_planes and eloss.infonce_small_enabled(B, T_, self.planes)
        if self.small:
            # (round 6) one process at the training batch size: the raw logits of all targets as ONE K-parallel plane GEMM, then two row-block kernels
            # (csrc/infonce_small.hip) -- the tile kernels below are 64 workgroups that each walk D alone (12 + 17 us of the critical path at B = 256)
            NC = T_ * B
            S_ = min(8, int(L.eegclip_head_gemm_slices(B, NC, Dm)))
            self.s_slabs = torch.empty(S_, B, NC, dtype=torch.float32, device=dev)
            self.s_ws = torch.empty(int(L.eegclip_infonce_small_workspace_floats(B, T_)), dtype=torch.float32, device=dev)
            pl.call_desc("eegclip_head_gemm", _abi.HeadGemmDesc(a_hi=self.q_planes[0].data_ptr(), a_lo=self.q_planes[1].data_ptr(), b_hi=self.t_planes[0].data_ptr(),
                                                               b_lo=self.t_planes[1].data_ptr(), lda=Dm, ldb=Dm, M=B, N=NC, K=Dm, slices=S_, slab_stride=B * NC,
                                                               C=self.s_slabs.data_ptr(), ldc=NC))
            pl.call("eegclip_infonce_small_fwd", self.s_slabs.data_ptr(), S_, B * NC, B, T_, sc_ptr, self.s_ws.data_ptr())
            w4 = list(self.weights) + [0.0] * (4 - T_)
            self.if_fwd_op, self.loss_arg = len(pl.ops), 11                 # (the op whose `loss` argument is patched per step)
            pl.call("eegclip_infonce_small_grad", B, T_, sc_ptr, self.s_ws.data_ptr(), *w4, self.Gp[0].data_ptr(), self.Gp[1].data_ptr(), NC, 0,
                    eng.G["logit_scale"].data_ptr())                        # d loss / d scale straight into its gradient
        # (loss.fused_infonce's training form: the forward leaves the per-tile partials, the gradient pass finalises them itself and adds the loss)
        elif os.environ.get("EEGCLIP_INFONCE_INLINE_FINALIZE", "1") != "0":
            pl.call("eegclip_infonce_fused_fwd", arr, 2 * T_, B, B, Dm, self.planes, B, sc_ptr, None)
            self.if_fwd_op, self.loss_arg = len(pl.ops), 8                  # (the op whose `loss` argument is patched per step)
            pl.call("eegclip_infonce_fused_grad_finalize", garr, T_, B, B, Dm, self.planes, B, sc_ptr, 0, eng.G["logit_scale"].data_ptr())      # d loss / d scale straight into its gradient
        else:
            for t_ in range(T_):
                garr[t_].part_k = garr[t_].diag_k = None
            self.if_fwd_op, self.loss_arg = len(pl.ops), 8
            pl.call("eegclip_infonce_fused_fwd", arr, 2 * T_, B, B, Dm, self.planes, B, sc_ptr, 0)
            pl.call("eegclip_infonce_fused_grad", garr, T_, B, B, Dm, self.planes, B, sc_ptr, eng.G["logit_scale"].data_ptr())
        # dA = [G_img | G_txt] [img; txt]: one launch, K = T B
        self.mse_op = None
        if self.on_planes:
            # ... K-parallel from planes (csrc/head_gemm.hip); the encoder's backward adds the slabs while its first kernel loads them
            S = int(L.eegclip_head_gemm_slices(B, Dm, T_ * B))
            self.da = torch.empty(S + (1 if self.mse_w else 0), B, Dm, dtype=torch.float32, device=dev)
            self.da_slices = S + (1 if self.mse_w else 0)
            pl.call_desc("eegclip_head_gemm", _abi.HeadGemmDesc(a_hi=self.Gp[0].data_ptr(), a_lo=self.Gp[1].data_ptr(), b_hi=self.t_planes[0].data_ptr(),
                                                               b_lo=self.t_planes[1].data_ptr(), lda=T_ * B, ldb=Dm, M=B, N=Dm, K=T_ * B, slices=S,
                                                               slab_stride=B * Dm, C=self.da.data_ptr(), ldc=Dm, b_kmajor=1))
            if self.mse_w:
                # the MSE term: weight * mean((z - img)^2) added to the step's loss, its gradient = the LAST slab of the backward's upstream gradient
                self.mse_op = len(pl.ops)
                pl.call("eegclip_mse_loss_grad_scaled", 0, 0, B * Dm, self.mse_w, 0, self.da[S].data_ptr())
        else:
            self.da = torch.empty(B, Dm, dtype=torch.float32, device=dev)
            self.da_slices = 1
            d = _abi.GemmDesc(M=B, N=Dm, K=2 * B, A=self.G.data_ptr(), Am=D(2 * B), Ak=D(1), B=self.stack.data_ptr(), Bk=D(Dm), Bn=D(1), C=self.da.data_ptr(),
                              Cm=D(Dm), Cn=D(1), Cpre=None, bias_n=None, bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=0, act=0, drop_p=0.0,
                              seed=0, drop_site=0, split_k=1, precision=_abi.PREC_BF16X3)
            pl._keep.append(d)
            pl.ops.append((L.eegclip_gemm_f32, [ctypes.byref(d), None], "eegclip_gemm_f32", False))
        # ---- encoder backward, op by op (self.bwd_index: backward-plan op -> this plan's op).  Inserted on the way: the accuracy readout behind the backward's
        # first second-stream launch (unless it forked behind the forward), and -- round 6 -- the optimizer update of the EARLY gradient bucket (loss scale,
        # conv stack, projection head: 82 % of the parameters) on the second stream as soon as the conv stack's backward has issued its last gradient, under
        # the transformer block's backward; the update at the end of the step then covers only the block's parameters (17 -> ~5 us on the exposed tail)
        self._plan_adamw(optimizer, eng)
        early_cut = getattr(self.bwd, "early_cut", None) if (self.adam_early and os.environ.get("EEGCLIP_ADAMW_EARLY", "1") != "0") else None
        taps_op = getattr(self.bwd, "taps_op", None)
        if early_cut is None or (taps_op is not None and taps_op < early_cut):
            early_cut = None
        b0 = len(pl.ops)
        self.bwd_base = b0
        self.bwd_index = [None] * len(self.bwd.ops)
        pl._seed_descs += self.bwd._seed_descs
        # (round 6, OPT-IN: EEGCLIP_WGRAD_ADAMW=1) the step ends  weight gradients of the block -> slab reduction -> [join] -> AdamW over what the early update
        # left; the reduction's sums can STEP THE OPTIMIZER themselves (eegclip_wgrad_tok_reduce_adamw: one launch for reduction + step + zero_grad, bit-identical).
        # Measured on the MI355X, alternated in one call: 0.6576 / 0.6589 ms per step against 0.6559 / 0.6566 with the two launches -- the fused launch takes
        # 15 us (scalar 4-byte accesses to P / M / V / G: the weight rows are 250 floats) against 11 + 6 and still waits for the join: not the default
        late = self.adam_late if early_cut is not None else self.adam_early + self.adam_late
        fuse_op = self._fusable_reduce(late) if os.environ.get("EEGCLIP_WGRAD_ADAMW", "0") == "1" else None

        def emit_bwd(i):
            fn, args, name, side = self.bwd.ops[i]
            self.bwd_index[i] = len(pl.ops)
            pl.ops.append((fn, list(args), name, side))
            pl._seed_slots += [(self.bwd_index[i], j) for i0, j in self.bwd._seed_slots if i0 == i]

        for i in range(len(self.bwd.ops)):
            if (early_cut is not None and i == taps_op) or i == fuse_op:
                continue                                     # (moved in front of the early update / behind the join, into the optimizer launch)
            emit_bwd(i)
            if i == self.bwd.dout_par_op and not self.acc_early:
                accuracy_ops()
            if early_cut is not None and i == early_cut - 1:
                if taps_op is not None:
                    emit_bwd(taps_op)
                self._emit_adamw(pl, self.adam_early, side=True)
        pl.set_arg(self.bwd_index[self.bwd.dout_op], 0, self.da.data_ptr())
        if self.on_planes:
            # the backward's first kernel adds the S slabs while it loads them and leaves the sum for the parameter half of the head's LayerNorm
            bb = eng.bufs[B]
            pl.set_arg(self.bwd_index[self.bwd.dout_op], 1, self.da_slices)
            pl.set_arg(self.bwd_index[self.bwd.dout_op], 2, B * Dm)
            pl.set_arg(self.bwd_index[self.bwd.dout_op], 13, bb["dout_sum"].data_ptr())
            pl.set_arg(self.bwd_index[self.bwd.dout_par_op], 0, bb["dout_sum"].data_ptr())
        else:
            pl.set_arg(self.bwd_index[self.bwd.dout_par_op], 0, self.da.data_ptr())
        pl.join()               # the optimizer reads every gradient (second-stream weight gradients) and rewrites logit_scale (read by the accuracy readout)
        # ---- fused AdamW + the zero_grad() that opens the next iteration (what the early update has not taken)
        if fuse_op is not None:
            fn, args, name, side = self.bwd.ops[fuse_op]
            (li, wp, gp, mp, vp, n), = late
            lr, b1, b2, eps, wd = self.hyper
            self.bwd_index[fuse_op] = len(pl.ops)
            self.adam_ops.append((len(pl.ops), li, 10))
            pl.call("eegclip_wgrad_tok_reduce_adamw", *args[:5], wp, gp, mp, vp, n, lr, b1, b2, eps, wd, 0)
        else:
            self._emit_adamw(pl, late)
        self.pl = pl
        self._class_ptr = None

    def _build_data_parallel(self, model, optimizer, eng, pl, splice, B, n_classes, Dm, L, dev):
        self.split_op = self.mse_op = None
        self.if_fwd_op = None
        self._cur = {}
        pl.callback(lambda: model.loss_func.gather_targets(self._cur["img"], self._cur["txt"]), "allgather_targets")
        f0 = splice(self.fwd)
        self.fwd_base = f0
        self.out_op = f0 + self.fwd.out_op
        sc_ptr = model.logit_scale.detach().reshape(1).data_ptr()
        ncp = (n_classes + 3) // 4 * 4
        self.q_planes = torch.empty(2, B, Dm, dtype=torch.bfloat16, device=dev)
        fn, args, name, side = pl.ops[self.out_op]
        self.acc_on_planes = name == "eegclip_residual_layernorm_fwd_slabs" and os.environ.get("EEGCLIP_HEAD_GEMM", "1") != "0"
        if self.acc_on_planes:
            args[19], args[20] = self.q_planes[0].data_ptr(), self.q_planes[1].data_ptr()
            self.logits = torch.empty(B, ncp, dtype=torch.float32, device=dev)
            self.class_planes = torch.zeros(2, ncp, Dm, dtype=torch.bfloat16, device=dev)
            self.acc_desc = None
            pl.call_desc("eegclip_head_gemm", _abi.HeadGemmDesc(a_hi=self.q_planes[0].data_ptr(), a_lo=self.q_planes[1].data_ptr(), b_hi=self.class_planes[0].data_ptr(),
                                                               b_lo=self.class_planes[1].data_ptr(), lda=Dm, ldb=Dm, M=B, N=ncp, K=Dm, slices=1, slab_stride=0,
                                                               C=self.logits.data_ptr(), ldc=ncp), side=True)
        else:
            self.logits = torch.empty(B, n_classes, dtype=torch.float32, device=dev)
            self.acc_desc = _abi.GemmDesc(M=B, N=n_classes, K=Dm, A=0, Am=D(Dm), Ak=D(1), B=0, Bk=D(1), Bn=D(Dm), C=self.logits.data_ptr(), Cm=D(n_classes), Cn=D(1),
                                          Cpre=None, bias_n=None, bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=0, act=0, drop_p=0.0, seed=0,
                                          drop_site=0, split_k=1, precision=self.fwd.precision)
            pl._keep.append(self.acc_desc)
            pl.ops.append((L.eegclip_gemm_f32, [ctypes.byref(self.acc_desc), None], "eegclip_gemm_f32", True))
        self.count_op = len(pl.ops)
        pl.call("eegclip_top1_count", self.logits.data_ptr(), B, n_classes, self.logits.shape[1], sc_ptr, 0, 0, side=True)
        self.acc_early = True
        # ---- the loss between the collectives
        b0_holder = {}

        def loss_cb():
            import types
            from .loss import _ClipLossFn
            cur = self._cur
            ctx = types.SimpleNamespace()
            a = cur["out"].requires_grad_(True)                  # (a fresh leaf: only its requires_grad flag is read)
            cur["loss"] = _ClipLossFn.forward(ctx, a, model.logit_scale, model.loss_func, self.weights, cur["img"], cur["txt"])
            da, ds, _ = ctx.grads
            cur["da"] = da                                        # (alive until the backward has been enqueued)
            eng.G["logit_scale"].copy_(ds.reshape(eng.G["logit_scale"].shape))
            pl.set_arg(b0_holder["b0"] + self.bwd.dout_op, 0, da.data_ptr())
            pl.set_arg(b0_holder["b0"] + self.bwd.dout_par_op, 0, da.data_ptr())
        pl.callback(loss_cb, "clip_loss_data_parallel")
        # ---- encoder backward (its callbacks: SyncBN sums, the early bucket's asynchronous all-reduce)
        b0 = len(pl.ops)
        b0_holder["b0"] = b0
        self.bwd_base = b0
        self.bwd_index = [b0 + i for i in range(len(self.bwd.ops))]
        splice(self.bwd)
        from . import dist as edist
        pl.callback(lambda: edist.average_flat_grads(eng.gflat, eng), "allreduce_flat_gradient")
        pl.join()
        self._plan_adamw(optimizer, eng)
        self._emit_adamw(pl, self.adam_early + self.adam_late)
        self.pl = pl
        self._class_ptr = None
        self.items, self.item0 = None, 0

    def _plan_adamw(self, optimizer, eng):
        """the optimizer's cached launches of this step (optim.AdamW: one per contiguous run of live parameters), each cut at the end of the engine's EARLY
        gradient bucket: self.adam_early / self.adam_late = [(launch index, weights, grads, m, v, count)]"""
        fast = optimizer._fast_last.get(0)
        self.fast = fast
        self.group = optimizer.param_groups[0]
        self.adam_ops = []                # (plan op, index into fast["launch"], argument index of lr): hyper-parameters at lr .. lr + 4, the run's step count behind them -- patched per call
        self.adam_early, self.adam_late = [], []
        base, (a0, a1) = eng.flat.data_ptr(), eng.early_bucket
        for li, (p0, n, wp, gp, mp, vp, members) in enumerate(fast["launch"]):
            off = (wp - base) // 4
            if not (0 <= off < eng.flat.numel()) or gp - eng.gflat.data_ptr() != wp - base:
                self.adam_late.append((li, wp, gp, mp, vp, n))                       # (not a run of this engine's flat buffers)
                continue
            n_early = max(0, min(off + n, a1) - max(off, a0)) if off <= a0 else 0    # (a run that starts inside the bucket is not cut)
            if off == a0 and n_early > 0:
                self.adam_early.append((li, wp, gp, mp, vp, n_early))
                if n > n_early:
                    self.adam_late.append((li, wp + 4 * n_early, gp + 4 * n_early, mp + 4 * n_early, vp + 4 * n_early, n - n_early))
            else:
                self.adam_late.append((li, wp, gp, mp, vp, n))
        g = self.group
        self.hyper = (g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])

    def _fusable_reduce(self, late):
        """index (in the backward plan) of the slab reduction of the block's weight gradients if its sums can step the optimizer directly: the last
        main-stream eegclip_wgrad_tok_reduce, ONE optimizer run left for the end of the step, every gradient of the launch dense inside that run"""
        if self.world != 1 or len(late) != 1:
            return None
        cand = [i for i, (fn, args, name, side) in enumerate(self.bwd.ops) if name == "eegclip_wgrad_tok_reduce" and not side]
        if not cand or any(name.startswith("eegclip_wgrad_tok") and i > cand[-1] for i, (fn, args, name, side) in enumerate(self.bwd.ops)):
            return None
        i = cand[-1]
        args = self.bwd.ops[i][1]
        arr, n_prob = args[0], int(args[1])
        li, wp, gp, mp, vp, n = late[0]
        for k in range(n_prob):
            q = arr[k]
            if q.ldo != q.N or q.sample_index or not (gp <= (q.out or 0) and q.out + 4 * q.M * q.N <= gp + 4 * n):
                return None
            if q.bias_out and not (gp <= q.bias_out and q.bias_out + 4 * q.M <= gp + 4 * n):
                return None
        return i

    def _emit_adamw(self, pl, launches, side=False):
        gs = None
        lr, b1, b2, eps, wd = self.hyper
        for li, wp, gp, mp, vp, n in launches:
            self.adam_ops.append((len(pl.ops), li, 5))
            pl.call("eegclip_adamw_step_zero_grad", wp, gp, mp, vp, n, lr, b1, b2, eps, wd, 0, 1.0, gs, side=side)

    def index_of(self, kind, i):
        """plan op index of op `i` of the spliced forward ("f") / backward ("b") plan (bench.py maps its per-launch timings through this)"""
        if kind == "f":
            return self.fwd_base + i
        return self.bwd_index[i]

    # ------------------------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def eligible(model, optimizer, eeg_data, subject_id, img, txt, labels, class_feats, objective, keep_grads, world):
        """can this call go through a step plan at all (cheap checks, every step)"""
        from .atms import ATMS
        from .loss import ClipLoss, fused_enabled, head_gemm_enabled, infonce_small_enabled
        from .optim import AdamW
        if not (enabled() and objective in ("retrieval", "reconstruction") and not keep_grads and _runtime_ok()):
            return False
        from .plan import default_gemm_precision
        if default_gemm_precision() != _abi.PREC_BF16X3:          # exact-fp32 plans are launch-per-Linear: nothing to build a plan from (and nothing to warn about)
            return False
        if world > 1 and (objective != "retrieval" or model.joint_train or os.environ.get("EEGCLIP_STEP_PLAN_DP", "1") == "0"):
            return False
        if not (isinstance(model, ATMS) and model.training and (isinstance(subject_id, int) or model.joint_train)):
            return False
        lf = model.loss_func
        if type(lf) is not ClipLoss or lf.world_size != world:
            return False
        if not isinstance(optimizer, AdamW) or len(optimizer.param_groups) != 1 or optimizer.grad_scale_dev is not None:
            return False
        B = eeg_data.shape[0]
        for t_, shape in ((eeg_data, None), (img, (B, 1024)), (class_feats, None)) + (((txt, (B, 1024)),) if objective == "retrieval" else ()):
            if not (_on_device(t_) and t_.dtype == torch.float32 and t_.is_contiguous() and not t_.requires_grad):
                return False
            if shape is not None and tuple(t_.shape) != shape:
                return False
        if labels.dtype != torch.long or not _on_device(labels) or labels.numel() != B or class_feats.dim() != 2 or class_feats.shape[1] != 1024:
            return False
        if world > 1 or fused_enabled(B, B, 1024):
            return True
        T_ = 2 if objective == "retrieval" else 1              # batches that are not whole 64-tiles: the small InfoNCE form only
        return bool(head_gemm_enabled(B, T_ * B, 1024) and infonce_small_enabled(B, T_, 2) and lf.logits_dtype == "f32")

    def still_valid(self, model, optimizer):
        """the captured state is still the live one: same engine / plans / optimizer launch cache, nobody attached gradients or changed hyper-parameters"""
        eng = self.eng
        if eng is None or model._eng is not eng or eng.stale(model) or eng.plans.get(("f",) + self.key) is not self.fwd or eng.plans.get(self.bwd_key) is not self.bwd:
            return False
        if self._model_ref() is not model or self._opt_ref() is not optimizer:
            return False
        if model.drop_probs(True) != self.probs or not optimizer.activate_launch_set(0, self.fast):
            return False
        g = self.group
        if optimizer.param_groups[0] is not g or (g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]) != self.hyper:
            b1, b2 = g["betas"]
            if optimizer.param_groups[0] is not g:
                return False
            self.hyper = (g["lr"], b1, b2, g["eps"], g["weight_decay"])       # (a learning-rate schedule: patch the optimizer launches)
            for op, _li, base in self.adam_ops:
                for j, v in enumerate(self.hyper):
                    self.pl.set_arg(op, base + j, v)
        return all(p.grad is None for p in g["params"])

    def run(self, eeg_data, img, txt, labels, class_feats, correct, subject_id=None):
        """enqueue one step; returns (features (B,1024), loss scalar) -- both device tensors, no host sync"""
        from .loss import _zero_pair
        eng, pl, B = self.eng, self.pl, self.B
        b = eng.bufs[B]
        if self.joint:
            # joint-subject model: the per-step subject layout (Embed.py:142-144) -- subject ids + the subject-ordered sample list go up in one asynchronous
            # copy, the per-subject weight-gradient problems of the backward are re-pointed (atms._Engine._joint_layout, exactly as the ordinary path);
            # the two arguments it patches on the backward plan's launches are mirrored onto this plan's copies
            import numpy as np
            from .retrieval import _uniform_ids
            ids = _uniform_ids(B, subject_id, eeg_data.device)
            host = getattr(ids, "_eegclip_host_ids", None)
            if host is None:
                host = np.full(B, int(subject_id), dtype=np.int64)
            uid = getattr(ids, "_eegclip_uniform_id", None)
            if uid is None or b.get("ids_uniform") != uid:
                b["ids"].copy_(ids)
                b["ids_uniform"] = uid
            eng._joint_layout(self.fwd, b, B, host, eeg_data.data_ptr(), False)
            eng._joint_layout(self.bwd, b, B, None, eeg_data.data_ptr(), True)
            for op in self.bwd.j_wk_ops:
                pl.set_arg(self.index_of("b", op), 1, self.bwd.ops[op][1][1])
                pl.set_arg(self.index_of("b", op), 3, self.bwd.ops[op][1][3])
        self.fwd.tb_desc.x = eeg_data.data_ptr()
        out = torch.empty(B, 1024, dtype=torch.float32, device=self.dev)
        op = out.data_ptr()
        pl.set_arg(self.out_op, 8, op)
        cp = (class_feats.data_ptr(), class_feats._version)
        if self.acc_on_planes:
            if cp != self._class_ptr:                  # a new (or modified) class table: its planes, once
                nc = class_feats.shape[0]
                it = (_abi.SplitItem * 1)(_abi.SplitItem(src=class_feats.data_ptr(), hi=self.class_planes[0].data_ptr(), lo=self.class_planes[1].data_ptr(), rows=nc,
                                                          cols=1024, ld_src=1024, ld_out=1024, transpose=0))
                rc = lib().eegclip_split_rows(it, 1, raw_stream())
                if rc:
                    raise RuntimeError(f"eegclip_split_rows returned {rc}")
                self._class_ptr = cp
        else:
            self.acc_desc.A = op
            if cp != self._class_ptr:
                self.acc_desc.B = cp[0]
                self._class_ptr = cp
        pl.set_arg(self.count_op, 5, labels.data_ptr())
        pl.set_arg(self.count_op, 6, correct.data_ptr())
        if self.world > 1:
            self._cur = {"img": img, "txt": txt, "out": out}
        else:
            self.items[self.item0].src = img.data_ptr()
            if self.T > 1:
                self.items[self.item0 + 1].src = txt.data_ptr()
        # the plan ACCUMULATES into the flat gradient buffer without attach_grads(): it must be clear.  It is when the optimizer's fused step cleared
        # exactly the views the last backward attached and nothing touched them since (every plan step leaves it so); after anything else -- a
        # keep_grads=True step, a manual backward followed by zero_grad(set_to_none=True): .grad is None but the buffer still holds values -- clear it here
        if eng._clear_for is None or eng._clear_for != eng._attached:
            eng.gflat.zero_()
        acc = _zero_pair(self.dev) if self.world == 1 else None
        if self.if_fwd_op is not None:
            pl.set_arg(self.if_fwd_op, self.loss_arg, acc.data_ptr())
        if self.mse_op is not None:
            pl.set_arg(self.mse_op, 0, op)
            pl.set_arg(self.mse_op, 1, img.data_ptr())
            pl.set_arg(self.mse_op, 4, acc.data_ptr())
        fast = self.fast
        fast["run_steps"] = [st + 1 for st in fast["run_steps"]]
        fast["pending"] += 1
        for opi, li, base in self.adam_ops:
            pl.set_arg(opi, base + 5, fast["run_steps"][li])
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if max(self.probs) > 0 else 0
        b["seed"] = seed
        pl._keep_step = (eeg_data, img, txt, labels, class_feats, out)      # (alive until the next step has been enqueued)
        pl.run(raw_stream(), seed)
        # what the launch-by-launch path leaves behind
        b["zb_clean"] = False
        eng.last_key = self.key
        eng.version[B] = eng.version.get(B, 0) + 1
        for own, ptrs in fast["owners"]:
            own.grads_cleared(ptrs)
        if self.world > 1:
            loss = self._cur.pop("loss")
            self._cur = {"da": self._cur.get("da")}
            return out, loss.reshape(())
        return out, acc[0].reshape(())
