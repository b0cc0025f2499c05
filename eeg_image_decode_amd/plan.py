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
from ._lib import check, lib

D = _abi.dim


def default_gemm_precision():
    """Arithmetic of the plan GEMMs (every Linear of the encoder / head / prior, forward and backward): split-bf16 products on the bf16 matrix
    cores by default (include/eegclip.h: EEGCLIP_PREC_BF16X3; embeddings move by <= 3e-5 against exact fp32 products, parity budget 1e-3);
    EEGCLIP_GEMM_PRECISION=f32 restores exact fp32 products (v_mfma_f32_16x16x4_f32, 1/16 of the rate)."""
    v = os.environ.get("EEGCLIP_GEMM_PRECISION", "bf16x3").lower()
    if v not in ("bf16x3", "f32"):
        raise ValueError("EEGCLIP_GEMM_PRECISION must be 'bf16x3' or 'f32'")
    return _abi.PREC_BF16X3 if v == "bf16x3" else _abi.PREC_F32


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
        self._side = None      # (torch side stream, {op index: fork event}, join event) -- created on first use
        self.use_side_stream = True
        self.skip = ()         # op indices left out of the next run()s (e.g. the value-embedding GEMMs of subjects absent from the batch)

    # -- generic positional op; `seed_at` = index of the seed argument (patched at run time)
    def call(self, fname, *args, seed_at=None, side=False):
        fn = getattr(self.L, fname)
        self.ops.append((fn, list(args) + [None], fname, side))
        if seed_at is not None:
            self._seed_slots.append((len(self.ops) - 1, seed_at))

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

    def memset(self, tensor):
        """zero a torch tensor as part of the plan (stream-ordered)"""
        self.ops.append((None, [tensor], "memset", False))

    def join(self):
        """the main stream waits for everything launched on the side stream so far"""
        self.ops.append((None, [None], "join", False))

    def op_names(self):
        return [op[2] for op in self.ops]

    def time_ops(self, indices):
        """record a HIP event pair around the given ops on every run (bench.py roofline / per-kernel breakdown)"""
        self.timed = {i: [] for i in indices}

    def timings_ms(self):
        import torch
        torch.cuda.synchronize()
        return {i: [a.elapsed_time(b) for a, b in evs] for i, evs in self.timed.items()}

    def run(self, stream, seed=0):
        for d in self._seed_descs:
            d.seed = seed
        for i, j in self._seed_slots:
            self.ops[i][1][j] = seed
        timed = self.timed
        import torch
        side = None
        if self.use_side_stream and torch.cuda.is_available() and any(op[3] for op in self.ops):
            if self._side is None:
                self._side = (torch.cuda.Stream(), {i: torch.cuda.Event() for i, op in enumerate(self.ops) if op[3]}, torch.cuda.Event())
            side = self._side
        ts = torch.cuda.current_stream() if (timed or side) else None
        dirty = False                                    # side stream has work the main stream has not waited for
        skip = self.skip
        for idx, (fn, args, name, on_side) in enumerate(self.ops):
            if skip and idx in skip:
                continue
            use_side = on_side and side is not None
            if timed and idx in timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side[0] if use_side else ts)
            if fn is None:
                if name == "memset":
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
                    with torch.cuda.stream(side[0]):
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
                if rc:
                    check(rc, f"{self.name}:{name}")
            if timed and idx in timed:
                e1.record(side[0] if use_side else ts)
                timed[idx].append((e0, e1))
        if dirty:
            side[2].record(side[0])
            ts.wait_event(side[2])
