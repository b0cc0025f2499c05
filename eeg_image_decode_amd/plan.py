"""A Plan is a pre-built, replayable list of C-ABI kernel launches (one training-step phase at one batch size).

Building the ctypes argument records costs tens of microseconds per launch in Python, so they are built ONCE for a
given shape; replaying a plan is a tight loop of foreign calls on the current HIP stream (and is what gets captured into
a hipGraph by `torch.cuda.graph`).  Only the dropout seed changes between replays: ops that draw a mask register a
seed slot that `run()` patches.

Side stream: an op built with ``side=True`` is launched on a second HIP stream after an event that orders it behind everything
enqueued so far on the main stream; the main stream does NOT wait for it until ``join()`` (implicit at the end of the plan).  The
backward plans use it for the weight-gradient kernels, whose results nobody reads before the optimizer step: a dW GEMM (512 workgroups
of 16 k-tiles) and the dX GEMM that follows it on the main stream (1024 workgroups of 8 k-tiles) then share the 256 CUs instead
of running as two half-empty waves of lock-stepped workgroups.
"""
import ctypes
import os

from . import _abi
from ._lib import check, cuda_available, current_stream, lib, use_stream

D = _abi.dim


# split-K through per-slice slabs + an ordered reduction instead of atomics on C: run-to-run bit-reproducible weight gradients.  Measured equal in
# time at the step's split (256 x 250 x 16384 / 32 slices: 31.4 us atomics, 32.3 us slabs; tools/wgrad_probe.py), so it is opt-in.
_SPLITK_WORKSPACE = False


_SIDE_PRIORITY = 0                                                    # HIP stream priority of a plan's second stream (range on this stack: 0 .. -1; round 5: -1 = 1.8 ms per step instead of 0.75)


def default_gemm_precision():
    """Arithmetic of the plan GEMMs (every Linear of the encoder / head / prior, forward and backward): split-bf16 products on the bf16 matrix
    cores by default (include/eegclip.h: EEGCLIP_PREC_BF16X3; embeddings move by <= 3e-5 against exact fp32 products, parity budget 1e-3);
    EEGCLIP_GEMM_PRECISION=f32 restores exact fp32 products (v_mfma_f32_16x16x4_f32, 1/16 of the rate)."""
    v = os.environ.get("EEGCLIP_GEMM_PRECISION", "bf16x3").lower()
    if v not in ("bf16x3", "f32"):
        raise ValueError("EEGCLIP_GEMM_PRECISION must be 'bf16x3' or 'f32'")
    return _abi.PREC_BF16X3 if v == "bf16x3" else _abi.PREC_F32


# entry points that launch exactly ONE kernel: eegclip_time_next_launch stamps the first kernel of a call, so only these take kernel timestamps
_SINGLE_KERNEL_OPS = frozenset({"eegclip_gemm_f32", "eegclip_token_block_fwd", "eegclip_token_block_bwd", "eegclip_attention_fwd", "eegclip_attention_bwd", "eegclip_attention_bwd_x3",
                                "eegclip_wgrad_tok", "eegclip_wgrad_tok_reduce"})


class Plan:
    def __init__(self, name="", precision=None):
        self.name = name
        self.precision = default_gemm_precision() if precision is None else precision
        self.ops = []          # [fn, [args..., stream]]
        self._keep = []        # keep ctypes structs / tensors alive
        self._seed_descs = []
        self._seed_slots = []  # (op index, arg index)
        self.L = lib()
        self.timed = {}        # op index -> list of (start, end) torch.cuda.Event pairs (HIP events on the launch stream)
        self.time_every, self._time_tick = 1, 0
        self.kernel_timestamps = os.environ.get("EEGCLIP_TIMING", "kernel") != "bracket"      # per-op timing: kernel timestamps | HIP-event brackets
        self._ev_free, self._ev_used = [], []
        self._side = None      # (torch side stream, {op index: fork event}, join event) -- created on first use
        self._side2 = None     # second side stream (ops flagged side=2), C executor only
        self.use_side_stream = True
        self.skip = ()         # op indices left out of the next run()s (e.g. the value-embedding GEMMs of subjects absent from the batch)
        self._c = None         # compiled form for the C executor (eegclip_plan_run): built on the first untimed run
        self._c_skip = ()
        self.use_c_executor = os.environ.get("EEGCLIP_PLAN_EXEC", "c") != "python"

    # -- generic positional op; `seed_at` = index of the seed argument (patched at run time)
    def call(self, fname, *args, seed_at=None, side=False):
        fn = getattr(self.L, fname)
        self.ops.append((fn, list(args) + [None], fname, side))
        if seed_at is not None:
            self._seed_slots.append((len(self.ops) - 1, seed_at))

    def call_desc(self, fname, d, seeded=False, side=False):
        """an op whose only argument is a descriptor struct owned by the plan (`seeded`: its .seed field takes the run's dropout seed)"""
        self._keep.append(d)
        if seeded:
            self._seed_descs.append(d)
        self.ops.append((getattr(self.L, fname), [ctypes.byref(d), None], fname, side))
        return d

    def desc(self, M, N, K, A, Am, Ak, B, Bk, Bn, C, Cm, Cn, *, Cpre=None, bias_n=None, bias_m=None, R=None, Rm=None, Rn=None,
             alpha=1.0, accumulate=0, act=0, drop_p=0.0, drop_site=0, split_k=1, rowsum_a=None, precision=None):
        """a GEMM descriptor owned by the plan but not (yet) an op: member template of a grouped launch"""
        d = _abi.GemmDesc(M=M, N=N, K=K, A=A, Am=Am, Ak=Ak, B=B, Bk=Bk, Bn=Bn, C=C, Cm=Cm, Cn=Cn, Cpre=Cpre, bias_n=bias_n,
                          bias_m=bias_m, R=R, Rm=Rm or D(0), Rn=Rn or D(0), alpha=alpha, accumulate=accumulate, act=act,
                          drop_p=drop_p, seed=0, drop_site=drop_site, split_k=split_k, rowsum_a=rowsum_a,
                          precision=self.precision if precision is None else precision)
        self._keep.append(d)
        return d

    def gemm(self, *a, side=False, **k):
        d = self.desc(*a, **k)
        if d.split_k > 1 and _SPLITK_WORKSPACE:
            # split-K through a scratch buffer of this op's own (ops of one plan may overlap on its two streams): partial tiles + one ordered
            # reduction instead of split_k atomic adds per output element
            nbytes = int(self.L.eegclip_gemm_workspace_bytes(ctypes.byref(d)))
            if nbytes > 0:
                import torch
                ws = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda" if torch.cuda.is_available() else "cpu")   # (cpu: the test emulator)
                self._keep.append(ws)
                d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
        if d.drop_p > 0.0:
            self._seed_descs.append(d)
        self.ops.append((self.L.eegclip_gemm_f32, [ctypes.byref(d), None], "eegclip_gemm_f32", side))
        return d

    def gemm_grouped(self, n_max, side=False):
        """one eegclip_gemm_f32_grouped op over a descriptor array the caller fills before each run: returns (array, op index); the member
        count is op argument 1"""
        arr = (_abi.GemmDesc * n_max)()
        self._keep.append(arr)
        self.ops.append((self.L.eegclip_gemm_f32_grouped, [arr, 0, None], "eegclip_gemm_f32_grouped", side))
        return arr, len(self.ops) - 1

    def callback(self, fn, name="callback", side=False):
        """run a host callable in stream order (collectives between kernels: SyncBN statistics).  side=True: the callable runs with the plan's
        second stream current, behind everything enqueued on both streams so far (an asynchronous collective over gradients that kernels of
        either stream have produced)"""
        self.ops.append((None, [fn], name, side))

    def memset(self, tensor, side=False):
        """zero a torch tensor as part of the plan (stream-ordered).  side=True: on the second stream, behind everything enqueued on the main stream so
        far -- the main stream sees the cleared buffer after the next join()"""
        self.ops.append((None, [tensor], "memset", bool(side)))

    def join(self):
        """the main stream waits for everything launched on the side stream so far"""
        self.ops.append((None, [None], "join", False))

    def op_names(self):
        return [op[2] for op in self.ops]

    def set_arg(self, idx, j, value):
        """patch argument j of op idx for the following runs (the per-call pointers: input batch, output tensor, upstream gradient)"""
        self.ops[idx][1][j] = value
        if self._c is not None:
            self._c_write(idx, j, value)

    # ---- C executor -----------------------------------------------------------------------------------------------------------------
    def _c_write(self, idx, j, value):
        arr, slots = self._c["arr"], self._c["slots"]
        if slots[idx] is None or j >= len(slots[idx]):
            return
        setattr(arr[idx].a[j], slots[idx][j], 0 if value is None else value)

    def _compile(self):
        """the op list as an eegclip_plan_op array: replayed by ONE foreign call (eegclip_plan_run) instead of one ctypes call per launch --
        the Python loop costs ~5 us per op, more than most of these kernels run for.  Host callbacks (collectives under data parallelism) cut the
        array into segments that are run one call each."""
        L = self.L
        fns = _abi.plan_functions()
        n = len(self.ops)
        arr = (_abi.PlanOp * n)()
        slots, segments, start = [], [], 0
        for i, (fn, args, name, on_side) in enumerate(self.ops):
            if fn is None and name == "memset":
                t = args[0]
                arr[i].fn = _abi.PLAN_MEMSET
                arr[i].flags = _abi.PLAN_SIDE if on_side else 0
                arr[i].a[0].p, arr[i].a[1].i = t.data_ptr(), t.numel() * t.element_size()
                slots.append(None)
            elif fn is None and name == "join":
                arr[i].fn = _abi.PLAN_JOIN
                slots.append(None)
            elif fn is None:                                      # host callback: a segment boundary
                arr[i].fn = -1
                slots.append(None)
                if i > start:
                    segments.append(("c", start, i))
                segments.append(("py", i, on_side))
                start = i + 1
            else:
                fid = L.eegclip_plan_fn_id(name.encode())
                if fid < 0:
                    raise RuntimeError(f"{name} is not dispatchable by the plan executor")
                arr[i].fn = fid
                arr[i].flags = (_abi.PLAN_SIDE2 if on_side == 2 else _abi.PLAN_SIDE) if on_side else 0
                sl = [_abi.plan_slot(t) for t in fns[name]]
                slots.append(sl)
                for j, s_ in enumerate(sl):
                    v = args[j]
                    if hasattr(v, "_obj"):                        # ctypes.byref(struct): the struct is owned by the plan (self._keep)
                        v = ctypes.addressof(v._obj)
                    elif isinstance(v, ctypes.Array):
                        v = ctypes.addressof(v)
                    setattr(arr[i].a[j], s_, 0 if v is None else v)
        if start < n:
            segments.append(("c", start, n))
        n_side = sum(1 for op in self.ops if op[3])
        ev = (ctypes.c_void_p * n)()
        join, join2 = ctypes.c_void_p(), ctypes.c_void_p()
        if n_side:
            tmp = (ctypes.c_void_p * (n_side + 2))()
            check(L.eegclip_plan_events(n_side + 2, tmp), f"{self.name}:plan_events")
            k = 0
            for i, op in enumerate(self.ops):
                if op[3]:
                    ev[i] = tmp[k]
                    k += 1
            join, join2 = ctypes.c_void_p(tmp[n_side]), ctypes.c_void_p(tmp[n_side + 1])
        self._c = dict(arr=arr, slots=slots, segments=segments, events=ev, join=join, join2=join2, n=n, dirty=ctypes.c_int(0), failed=ctypes.c_int(-1),
                       owned=(tmp if n_side else None))
        self._c_skip = ()

    def close(self):
        """give the plan's HIP events back (fork / join events of the C executor, library-owned timing events); the plan must not run afterwards.
        Called when an engine drops a plan and from __del__: plans are rebuilt per (batch size, mode, world size ...) key."""
        c, self._c = self._c, None
        try:
            if c is not None and c.get("owned") is not None:
                self.L.eegclip_plan_events_destroy(len(c["owned"]), c["owned"])
            for e in self._ev_free + self._ev_used:
                self.L.eegclip_timing_event_destroy(e)
        except Exception:                                         # interpreter shutdown: the library may be gone
            pass
        self._ev_free, self._ev_used = [], []

    def __del__(self):
        self.close()

    def _run_c(self, stream, seed):
        import torch
        c = self._c
        arr = c["arr"]
        for i, j in self._seed_slots:
            arr[i].a[j].u = seed
        if self.skip != self._c_skip:
            for i in self._c_skip:
                arr[i].flags &= ~_abi.PLAN_SKIP
            for i in self.skip:
                arr[i].flags |= _abi.PLAN_SKIP
            self._c_skip = self.skip
        side = side2 = None
        if self.use_side_stream and cuda_available() and self._uses_side():
            if self._side is None:
                self._side = (torch.cuda.Stream(priority=_SIDE_PRIORITY), None, None)
            side = self._side[0]
            if self._side_kinds[1]:                               # a second side stream: independent weight-gradient GEMMs side by side
                if self._side2 is None:
                    self._side2 = torch.cuda.Stream(priority=_SIDE_PRIORITY)
                side2 = self._side2
        c["dirty"].value = 0
        for seg in c["segments"]:
            if seg[0] == "c":
                rc = self.L.eegclip_plan_run(arr, seg[1], seg[2], c["n"], stream, side.cuda_stream if side is not None else None,
                                             side2.cuda_stream if side2 is not None else None, c["events"], c["join"], c["join2"],
                                             ctypes.byref(c["dirty"]), ctypes.byref(c["failed"]))
                if rc:
                    check(rc, f"{self.name}:{self.ops[c['failed'].value][2]}")
            else:
                idx, on_side = seg[1], seg[2]
                if idx in self.skip:
                    continue
                fn = self.ops[idx][1][0]
                if on_side and side is not None:                  # behind everything enqueued on ALL streams so far
                    ts = current_stream()
                    e = torch.cuda.Event()
                    e.record(ts)
                    side.wait_event(e)
                    if side2 is not None and (c["dirty"].value & 2):
                        e2 = torch.cuda.Event()
                        e2.record(side2)
                        side.wait_event(e2)
                    with use_stream(side):
                        fn()
                    c["dirty"].value |= 1
                else:
                    fn()
        if c["dirty"].value and side is not None:                 # (a trailing side-stream callback: join here)
            e = torch.cuda.Event()
            e.record(side)
            current_stream().wait_event(e)
            if side2 is not None and (c["dirty"].value & 2):
                e2 = torch.cuda.Event()
                e2.record(side2)
                current_stream().wait_event(e2)
            c["dirty"].value = 0

    def _uses_side(self):
        """does any op run on a side stream (and on the second one)?  Asked per run: cached per op count (plans only grow while they are built)"""
        k = getattr(self, "_side_kinds", None)
        if k is None or k[2] != len(self.ops):
            k = self._side_kinds = (any(op[3] for op in self.ops), any(op[3] == 2 for op in self.ops), len(self.ops))
        return k[0]

    def time_ops(self, indices, every=1):
        """record a HIP event pair around the given ops on every `every`-th run (bench.py roofline / per-kernel breakdown); the other runs go
        through the normal (un-instrumented) path"""
        self.timed = {i: [] for i in indices}
        self.time_every, self._time_tick = max(1, int(every)), 0
        self._ev_free.extend(self._ev_used)          # (the previous measurement has been read)
        self._ev_used = []

    def _timing_event(self):
        """a library-owned event for eegclip_time_next_launch (recycled across time_ops() calls)"""
        if self._ev_free:
            e = self._ev_free.pop()
        else:
            e = self.L.eegclip_timing_event_create()
            if not e:
                raise RuntimeError("hipEventCreate failed")
        self._ev_used.append(e)
        return e

    def timings_ms(self):
        import torch
        torch.cuda.synchronize()
        el = self.L.eegclip_timing_elapsed_ms
        return {i: [(float(el(a, b)) if isinstance(a, int) else a.elapsed_time(b)) for a, b in evs] for i, evs in self.timed.items()}

    def run(self, stream, seed=0):
        for d in self._seed_descs:
            d.seed = seed
        timed = self.timed
        if timed and self.time_every > 1:
            self._time_tick += 1
            if self._time_tick % self.time_every:
                timed = None
        if self.use_c_executor and not timed:
            if self._c is None:
                self._compile()
            return self._run_c(stream, seed)
        for i, j in self._seed_slots:
            self.ops[i][1][j] = seed
        import torch
        side = None
        if self.use_side_stream and cuda_available() and self._uses_side():
            if self._side is None or self._side[1] is None:        # (the C executor keeps only the stream: it owns its own events)
                st_side = self._side[0] if self._side is not None else torch.cuda.Stream(priority=_SIDE_PRIORITY)
                self._side = (st_side, {i: torch.cuda.Event() for i, op in enumerate(self.ops) if op[3]}, torch.cuda.Event())
            side = self._side
        ts = current_stream() if (timed or side) else None
        dirty = False                                    # side stream has work the main stream has not waited for
        skip = self.skip
        for idx, (fn, args, name, on_side) in enumerate(self.ops):
            if skip and idx in skip:
                continue
            use_side = on_side and side is not None
            # (single-kernel entry points only: the stamp covers the FIRST kernel a call launches)
            stamp = timed and idx in timed and name in _SINGLE_KERNEL_OPS and self.kernel_timestamps
            if stamp:
                # the kernel's own GPU begin / end timestamps (hipExtLaunchKernel events armed for the next launch): no marker packets around it
                e0, e1 = self._timing_event(), self._timing_event()
                self.L.eegclip_time_next_launch(e0, e1)
            elif timed and idx in timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side[0] if use_side else ts)
            if fn is None:
                if name == "memset" and use_side:
                    ev = side[1][idx]
                    ev.record(ts)
                    side[0].wait_event(ev)
                    with use_stream(side[0]):
                        args[0].zero_()
                    dirty = True
                elif name == "memset":
                    args[0].zero_()
                elif name == "join":
                    if dirty:
                        side[2].record(side[0])
                        ts.wait_event(side[2])
                        dirty = False
                elif use_side:
                    ev = side[1][idx]
                    ev.record(ts)
                    side[0].wait_event(ev)
                    with use_stream(side[0]):
                        args[0]()
                    dirty = True
                else:
                    args[0]()
            else:
                if use_side:
                    ev = side[1][idx]
                    ev.record(ts)                        # order the side op behind everything enqueued on the main stream so far
                    side[0].wait_event(ev)
                    args[-1] = side[0].cuda_stream
                    dirty = True
                else:
                    args[-1] = stream
                rc = fn(*args)
                if stamp:
                    self.L.eegclip_time_next_launch(None, None)      # (a call that returned before launching must not leave the pair armed)
                if rc:
                    check(rc, f"{self.name}:{name}")
            if stamp:
                timed[idx].append((e0, e1))
            elif timed and idx in timed:
                e1.record(side[0] if use_side else ts)
                timed[idx].append((e0, e1))
        if dirty:
            side[2].record(side[0])
            ts.wait_event(side[2])
