"""ATM-S EEG encoder on MI355X: same Python surface as the reference (Retrieval/ATMS_retrieval.py:171-191), all
arithmetic in hand-written HIP kernels reached through the C ABI (include/eegclip.h).

    model = ATMS().cuda();  z = model(eeg (B,63,250) f32, subject_ids (B,) i64)   ->  (B,1024) f32

* The module tree reproduces the reference ``state_dict()`` key for key, so reference
  ``.pth`` checkpoints load with ``load_state_dict`` and ours load into the reference.  The ``nn.Linear`` / ``nn.Conv2d``
  / ``nn.LayerNorm`` ... children are parameter holders with torch's default initialisers; they are never *called*.
* All parameters are views into ONE flat fp32 buffer (and their ``.grad`` into one flat gradient buffer): Q/K/V weights
  are adjacent so the three projections are a single GEMM, the optimizer step is one fused kernel launch per segment,
  and data-parallel gradient reduction is one RCCL all-reduce.
* forward/backward are replayed launch plans (plan.py) per batch size; activations live in persistent buffers, so the
  backward of a batch must run before the next forward at the same batch size (the training loop does exactly that).
* There is no CPU / eager-PyTorch fallback: a CPU tensor or a missing library raises.
"""
import os
import weakref
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

from . import _abi
from ._lib import EegclipError, current_stream, lib, raw_stream, require_cuda
from .loss import ClipLoss
from .plan import Plan

D = _abi.dim
ACT_GELU = _abi.ACT_GELU
ACT_GELU_GRAD = _abi.ACT_GELU_GRAD

N_CH, T_LEN, D_MODEL, N_HEADS, D_HEAD, D_FF = 63, 250, 250, 4, 62, 256
L_TOK = N_CH + 1
HE = N_HEADS * D_HEAD            # 248: d_keys = 250 // 4 (SelfAttention_Family.py:184)
C_TS, W_TS, F_TS, P_DIM = 40, 36, 1440, 1024
EPS = 1e-5
SITE_EMBED, SITE_ATTN, SITE_ATTN_OUT, SITE_FFN_ACT, SITE_FFN_OUT, SITE_CONV, SITE_PROJ = range(7)


class Config:
    """Hyper-parameters of the iTransformer front (Retrieval/ATMS_retrieval.py:44-59)."""

    def __init__(self):
        self.task_name = 'classification'
        self.seq_len = 250
        self.pred_len = 250
        self.output_attention = False
        self.d_model = 250
        self.embed = 'timeF'
        self.freq = 'h'
        self.dropout = 0.25
        self.factor = 1
        self.n_heads = 4
        self.e_layers = 1
        self.d_ff = 256
        self.activation = 'gelu'
        self.enc_in = 63


class _Holder(nn.Module):
    """Parameter container: its arithmetic runs inside ATMS.forward (HIP kernels), never on its own."""

    def forward(self, *a, **k):
        raise EegclipError(f"{type(self).__name__} holds parameters only; call ATMS.forward (HIP kernels). There is no eager path.")


class PositionalEmbedding(_Holder):
    def __init__(self, d_model, max_len=5000):
        super().__init__()
        pos = torch.arange(0, max_len).float().unsqueeze(1)
        div = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer('pe', pe.unsqueeze(0))


class TimeFeatureEmbedding(_Holder):
    def __init__(self, d_model):
        super().__init__()
        self.embed = nn.Linear(4, d_model, bias=False)      # dead parameter in the reference (x_mark is None)


class SubjectEmbedding(_Holder):
    def __init__(self, num_subjects, d_model):
        super().__init__()
        self.subject_embedding = nn.Embedding(num_subjects, d_model)
        self.shared_embedding = nn.Parameter(torch.randn(1, d_model))
        self.mask_embedding = nn.Parameter(torch.randn(1, d_model))


class DataEmbedding(_Holder):
    def __init__(self, c_in, d_model, dropout, num_subjects, joint_train=False):
        super().__init__()
        self.joint_train = bool(joint_train and num_subjects is not None)
        if self.joint_train:      # one value embedding per subject, chosen per sample (Embed.py:127-131,142-144)
            self.value_embedding = nn.ModuleDict({str(s): nn.Linear(c_in, d_model) for s in range(num_subjects)})
        else:
            self.value_embedding = nn.Linear(c_in, d_model)
        self.position_embedding = PositionalEmbedding(d_model)
        self.temporal_embedding = TimeFeatureEmbedding(d_model)
        self.dropout = nn.Dropout(p=dropout)
        self.subject_embedding = SubjectEmbedding(num_subjects, d_model)
        self.mask_token = nn.Parameter(torch.randn(1, d_model))


class FullAttention(_Holder):
    def __init__(self, attention_dropout):
        super().__init__()
        self.dropout = nn.Dropout(attention_dropout)


class AttentionLayer(_Holder):
    def __init__(self, attention, d_model, n_heads):
        super().__init__()
        dk = d_model // n_heads
        self.inner_attention = attention
        self.query_projection = nn.Linear(d_model, dk * n_heads)
        self.key_projection = nn.Linear(d_model, dk * n_heads)
        self.value_projection = nn.Linear(d_model, dk * n_heads)
        self.out_projection = nn.Linear(dk * n_heads, d_model)
        self.n_heads = n_heads


class EncoderLayer(_Holder):
    def __init__(self, attention, d_model, d_ff, dropout):
        super().__init__()
        self.attention = attention
        self.conv1 = nn.Conv1d(d_model, d_ff, 1)
        self.conv2 = nn.Conv1d(d_ff, d_model, 1)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)


class Encoder(_Holder):
    def __init__(self, attn_layers, norm_layer):
        super().__init__()
        self.attn_layers = nn.ModuleList(attn_layers)
        self.norm = norm_layer


class iTransformer(_Holder):
    def __init__(self, configs, joint_train=False, num_subjects=10):
        super().__init__()
        self.enc_embedding = DataEmbedding(configs.seq_len, configs.d_model, configs.dropout, num_subjects, joint_train)
        self.encoder = Encoder(
            [EncoderLayer(AttentionLayer(FullAttention(configs.dropout), configs.d_model, configs.n_heads),
                          configs.d_model, configs.d_ff, configs.dropout) for _ in range(configs.e_layers)],
            norm_layer=nn.LayerNorm(configs.d_model))


class PatchEmbedding(_Holder):
    def __init__(self, emb_size=40):
        super().__init__()
        self.tsconv = nn.Sequential(
            nn.Conv2d(1, 40, (1, 25), stride=(1, 1)), nn.AvgPool2d((1, 51), (1, 5)), nn.BatchNorm2d(40), nn.ELU(),
            nn.Conv2d(40, 40, (63, 1), stride=(1, 1)), nn.BatchNorm2d(40), nn.ELU(), nn.Dropout(0.5))
        self.projection = nn.Sequential(nn.Conv2d(40, emb_size, (1, 1), stride=(1, 1)), nn.Identity())


class FlattenHead(_Holder):
    pass


class Enc_eeg(nn.Sequential):
    def __init__(self, emb_size=40, **kwargs):
        super().__init__(PatchEmbedding(emb_size), FlattenHead())

    def forward(self, *a, **k):
        raise EegclipError("Enc_eeg holds parameters only; call ATMS.forward")


class ResidualAdd(_Holder):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class Proj_eeg(nn.Sequential):
    def __init__(self, embedding_dim=1440, proj_dim=1024, drop_proj=0.5):
        super().__init__(nn.Linear(embedding_dim, proj_dim),
                         ResidualAdd(nn.Sequential(nn.GELU(), nn.Linear(proj_dim, proj_dim), nn.Dropout(drop_proj))),
                         nn.LayerNorm(proj_dim))

    def forward(self, *a, **k):
        raise EegclipError("Proj_eeg holds parameters only; call ATMS.forward")


# flat-buffer order: the always-live group first (QKV adjacent), then the two conditionally-live token tensors, then dead
WGRAD_TOK_MAX_PROBLEMS = 12          # csrc/wgrad_tok.hip: WK_MAXP
_E, _LY, _TS = "encoder.enc_embedding.", "encoder.encoder.attn_layers.0.", "enc_eeg.0.tsconv."
_LIVE = [
    # gradients that are complete EARLY in the backward first (loss -> head -> conv stack: 10.5 of the 12.8 MB), the transformer's after them: under
    # data parallelism the flat gradient is then exactly TWO all-reduces, the first one started inside the backward plan (Engine.early_bucket)
    "logit_scale",
    _TS + "0.weight", _TS + "0.bias", _TS + "2.weight", _TS + "2.bias", _TS + "4.weight", _TS + "4.bias", _TS + "5.weight", _TS + "5.bias",
    "enc_eeg.0.projection.0.weight", "enc_eeg.0.projection.0.bias",
    "proj_eeg.0.weight", "proj_eeg.0.bias", "proj_eeg.1.fn.1.weight", "proj_eeg.1.fn.1.bias", "proj_eeg.2.weight", "proj_eeg.2.bias",
    _LY + "attention.query_projection.weight", _LY + "attention.key_projection.weight", _LY + "attention.value_projection.weight",
    _LY + "attention.query_projection.bias", _LY + "attention.key_projection.bias", _LY + "attention.value_projection.bias",
    _LY + "attention.out_projection.weight", _LY + "attention.out_projection.bias",
    _LY + "conv1.weight", _LY + "conv1.bias", _LY + "conv2.weight", _LY + "conv2.bias",
    _LY + "norm1.weight", _LY + "norm1.bias", _LY + "norm2.weight", _LY + "norm2.bias",
    "encoder.encoder.norm.weight", "encoder.encoder.norm.bias",
    # LAST: the gradient that completes last (its dY is the final output of the backward chain)
    _E + "value_embedding.weight", _E + "value_embedding.bias",
]
_CHECK_KEY = "encoder.encoder.norm.bias"          # (a live parameter of every model variant: Engine.stale)
_TOK_TABLE = _E + "subject_embedding.subject_embedding.weight"
_TOK_SHARED = _E + "subject_embedding.shared_embedding"


class ATMS(nn.Module):
    def __init__(self, num_channels=63, sequence_length=250, num_subjects=2, num_features=64, num_latents=1024, num_blocks=1, *,
                 joint_train=False, table_subjects=10):
        """Positional signature = Retrieval/ATMS_retrieval.py:170 (num_subjects only sizes the dead subject_wise_linear list there; the subject-token
        table always has 10 rows).  Keyword-only extensions for the joint-subject script (retrieval_joint.ATMS wraps them with that script's
        signature): joint_train = one value-embedding Linear per subject, table_subjects = rows of the subject-token table."""
        super().__init__()
        if num_channels != N_CH or sequence_length != T_LEN:
            raise EegclipError("the HIP kernels are specialised for 63 channels x 250 samples (the reference's only configuration)")
        cfg = Config()
        self.joint_train = bool(joint_train)
        self.table_subjects = int(table_subjects)
        self.encoder = iTransformer(cfg, joint_train, table_subjects)
        self.subject_wise_linear = nn.ModuleList([nn.Linear(cfg.d_model, sequence_length) for _ in range(num_subjects)])
        self.enc_eeg = Enc_eeg()
        self.proj_eeg = Proj_eeg()
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.loss_func = ClipLoss()
        self.sync_batchnorm = True      # under torch.distributed: BatchNorm batch statistics over the global batch (= the single-process semantics)
        self._eng = None

    # ---- flat parameter storage ------------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self._eng = None                     # .cuda()/.to()/.float() re-created the tensors: re-flatten lazily
        return r

    def _engine(self):
        p0 = self.logit_scale
        if self._eng is None or self._eng.stale(self):
            require_cuda(p0.data, "ATMS parameters (call model.cuda() first)")
            self._eng = _Engine(self)
        return self._eng

    def flat_parameters(self):
        """(flat weights, flat grads, segments) -- segments = [(offset, numel, keys)] live / table / shared / dead."""
        e = self._engine()
        return e.flat, e.gflat, e.segments

    # ---- reference API ---------------------------------------------------------------------------------------
    def forward(self, x, subject_ids):
        """x (B,63,250) f32 cuda, subject_ids (B,) int64 (or an int / None) -> (B,1024) f32.
        Any id >= 10 (e.g. 'sub-10'), or None, selects the shared token for the whole batch (Embed.py:116-119)."""
        eng = self._engine()
        require_cuda(x, "x")
        if x.dtype != torch.float32 or x.dim() != 3 or x.shape[1] != N_CH or x.shape[2] != T_LEN:
            raise EegclipError(f"x must be float32 (B,{N_CH},{T_LEN}); got {x.dtype} {tuple(x.shape)}")
        host_ids = None
        if self.joint_train:
            # one value embedding per subject: the subject of every sample must be known on the host (the reference reads subject_id.item() per
            # sample, Embed.py:144; an id without an embedding is a KeyError there).  Our loops attach the host copy they built the tensor from.
            if subject_ids is None:
                raise EegclipError("joint_train model: subject_ids is required (one value embedding per subject)")
            if isinstance(subject_ids, int):
                host_ids = [subject_ids] * x.shape[0]
                subject_ids = torch.full((x.shape[0],), subject_ids, dtype=torch.long, device=x.device)
            else:
                hint = getattr(subject_ids, "_eegclip_uniform_id", None)
                host_ids = getattr(subject_ids, "_eegclip_host_ids", None)
                if host_ids is None:
                    host_ids = [hint] * x.shape[0] if hint is not None else subject_ids.tolist()
            if len(host_ids) != x.shape[0]:
                raise EegclipError(f"subject_ids has {len(host_ids)} entries for a batch of {x.shape[0]}")
            host_ids = np.asarray(host_ids, dtype=np.int64)
            if host_ids.size and (host_ids.min() < 0 or host_ids.max() >= self.table_subjects):
                bad = host_ids[(host_ids < 0) | (host_ids >= self.table_subjects)]
                raise EegclipError(f"joint_train model has value embeddings for subjects 0..{self.table_subjects - 1}; got id {int(bad[0])}")
            ids, shared = subject_ids.to(device=x.device, dtype=torch.long), False
        elif subject_ids is None:
            ids, shared = None, True
        elif isinstance(subject_ids, int):
            shared = subject_ids >= 10
            ids = None if shared else torch.full((x.shape[0],), subject_ids, dtype=torch.long, device=x.device)
        else:
            ids = subject_ids.to(device=x.device, dtype=torch.long)
            hint = getattr(subject_ids, "_eegclip_uniform_id", None)        # set by our train/eval loops: no host sync
            host = getattr(subject_ids, "_eegclip_host_ids", None)          # per-sample ids our loops built the tensor from
            if hint is not None:
                shared = hint >= 10
            elif host is not None:
                shared = bool((np.asarray(host) >= 10).any())
            else:
                shared = bool((ids >= 10).any())                            # the reference syncs here too
            if shared:
                ids = None
        x = x.contiguous()
        train = self.training
        need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if need_grad:
            return _AtmsFn.apply(x, eng.anchor, self, ids, shared, train, host_ids)
        return eng.forward(x, ids, shared, train, host_ids)

    def drop_probs(self, train):
        if not train:
            return (0.0, 0.0, 0.0)
        e = self.encoder
        pe = {e.enc_embedding.dropout.p, e.encoder.attn_layers[0].attention.inner_attention.dropout.p, e.encoder.attn_layers[0].dropout.p}
        if len(pe) != 1:
            raise EegclipError("the three encoder dropout modules must share one p (the reference uses Config.dropout for all)")
        return (float(pe.pop()), float(self.enc_eeg[0].tsconv[7].p), float(self.proj_eeg[1].fn[2].p))


class _AtmsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, model, ids, shared, train, host_ids=None):
        eng = model._engine()
        out = eng.forward(x, ids, shared, train, host_ids)
        ctx.eng, ctx.key, ctx.version = eng, eng.last_key, eng.version[x.shape[0]]
        ctx.x = x
        ctx.want_dx = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        if eng.version.get(ctx.key[0]) != ctx.version:        # the buffers are shared by EVERY plan of this batch size (train / eval / token branch)
            raise EegclipError("ATMS activations were overwritten by a later forward at the same batch size; "
                               "run backward before the next forward (persistent activation buffers).")
        dx = eng.backward(ctx.key, ctx.x, dout.contiguous(), ctx.want_dx)
        return dx, None, None, None, None, None, None


def _p(t):
    return t.data_ptr()


def _dp_world():
    """ranks that share the parameters (1 without torch.distributed)"""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _head_split(B):
    """split-K factor of the projection-head GEMMs (M = B is small: a 4 x 16 tile grid cannot fill 256 CUs).  Every slice adds its
    partial tile with atomics, so the factor trades workgroups against B * 1024 * split float atomics."""
    import os
    cap = int(os.environ.get("EEGCLIP_HEAD_SK", "8"))        # tuning aid; measured at B = 256: 16 -> 101 us, 8 -> 80 us, 4 -> 103 us for the four GEMMs
    return max(1, min(cap, 2048 // max(1, ((B + 63) // 64) * (P_DIM // 64))))


class _Engine:
    """Flat parameter/gradient storage + per-batch-size activation buffers and launch plans."""
    check_cleared = False        # debug: verify the flat gradient buffer whenever attach_grads() skips its clear (see there)

    @property
    def model(self):
        m = self._model_ref()
        if m is None:
            raise EegclipError("the ATMS model of this engine has been garbage-collected")
        return m

    def __init__(self, model):
        sd_params = dict(model.named_parameters())
        dev = model.logit_scale.device
        self.device = dev
        self._model_ref = weakref.ref(model)          # (weak: model._eng -> engine is the only strong edge, so a dropped model frees its engine --
        #                                                 flat buffers, activation buffers, plans -- even after retrieval.settle_gc()'s gc.freeze())
        # joint-subject model: the single value embedding is replaced by one Linear per subject, each live only in steps whose batch holds
        # that subject (the reference never touches the others: their .grad stays None and AdamW skips them)
        self.joint = bool(model.joint_train)
        self.n_subj = model.table_subjects if self.joint else 0
        self.ve_keys = [(_E + f"value_embedding.{s}.weight", _E + f"value_embedding.{s}.bias") for s in range(self.n_subj)]
        if self.joint:
            self.live_base = [k for k in _LIVE if ".value_embedding." not in k]
            live = self.live_base + [k for pair in self.ve_keys for k in pair]
        else:
            live = self.live_base = list(_LIVE)
        dead = [k for k in sd_params if k not in live and k not in (_TOK_TABLE, _TOK_SHARED)]
        order = live + [_TOK_TABLE, _TOK_SHARED] + dead
        assert sorted(order) == sorted(sd_params), "parameter list drifted from the reference state_dict"
        offs, off = {}, 0
        for k in order:
            offs[k] = off
            off += (sd_params[k].numel() + 3) // 4 * 4          # 16-byte aligned segments
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)     # makes autograd call _AtmsFn.backward
        self._clear_for = self._attached = None          # see attach_grads / grads_cleared
        self._live_cache, self._attached_ptrs = {}, {}
        self.P, self.G, self.params = {}, {}, {}
        for k in order:
            p = sd_params[k]
            n = p.numel()
            view = self.flat[offs[k]:offs[k] + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            self.P[k] = view
            self.G[k] = self.gflat[offs[k]:offs[k] + n].view(p.shape)
            self.params[k] = p
            p._eegclip_grad_owner = weakref.ref(self)         # lets optim.AdamW.step(zero_grad=True) tell this engine that its buffer is clear
        n_live = offs[_TOK_TABLE]
        self.segments = [(0, n_live, list(live)),
                         (offs[_TOK_TABLE], sd_params[_TOK_TABLE].numel(), [_TOK_TABLE]),
                         (offs[_TOK_SHARED], sd_params[_TOK_SHARED].numel(), [_TOK_SHARED]),
                         (offs[dead[0]] if dead else off, off - (offs[dead[0]] if dead else off), dead)]
        self.offs = offs
        # gradient bucket that is complete early in the backward (conv stack + projection head: 10.5 of the 12.8 MB), contiguous in the flat
        # buffer; under data parallelism its all-reduce is started from inside the backward plan and hidden under the transformer's backward
        a0, a1 = 0, offs["proj_eeg.2.bias"] + (sd_params["proj_eeg.2.bias"].numel() + 3) // 4 * 4
        assert offs["logit_scale"] == 0 and a1 == offs[self.live_base[self.live_base.index("proj_eeg.2.bias") + 1]]
        self.early_bucket = (a0, a1)
        self.early_work = None
        self._check = (self.params["logit_scale"], self.params[_CHECK_KEY])
        self.buffers = dict(model.named_buffers())
        self.bufs, self.plans, self.version = {}, {}, {}
        self.last_key = None
        self.grad_fresh = True               # gflat holds zeros / stale values that must be cleared before accumulation
        lib()

    def _head_planes_enabled(self, pl):
        """the projection head (and, in the step plan / ClipLoss, the query gradient) on the K-parallel plane GEMM csrc/head_gemm.hip (round 6) in split-bf16
        plans; EEGCLIP_HEAD_GEMM=0 pins round 5's split-K gemm_x3 launches (diagnosis, A/B timing); exact-fp32 plans always use those"""
        return pl.precision == _abi.PREC_BF16X3 and os.environ.get("EEGCLIP_HEAD_GEMM", "1") != "0"

    def _head_weight_planes(self):
        """bf16 hi | lo planes of this step's projection-head weights: the span [W1 | b1 | W2] as it lies in the flat parameter buffer, ONE dense split.  The
        forward GEMMs read them as B (N, K) operands (k contiguous); the input-gradient GEMMs, which contract over the OUTPUT index, read the SAME planes as
        k-major operands (csrc/head_gemm.hip b_kmajor: fragments through the LDS transpose read) -- no transposed copy.  Returns the split item: it RIDES in
        the 1x1-conv launch in front of the head (csrc/split_rider.h: extra workgroups of a launch that leaves the memory system idle), so that the split
        costs no launch, no second stream and no join."""
        P = self.P
        w1, w2 = P["proj_eeg.0.weight"], P["proj_eeg.1.fn.1.weight"]
        span = (w2.data_ptr() - w1.data_ptr()) // 4 + w2.numel()
        assert w2.data_ptr() > w1.data_ptr() and span % 8 == 0 and span < 4 * w1.numel(), "projection-head weights are not adjacent in the flat buffer"
        if not hasattr(self, "hw_planes"):
            self.hw_planes = torch.empty(2, span, dtype=torch.bfloat16, device=self.device)
        o2 = (w2.data_ptr() - w1.data_ptr()) // 4
        # (hi, lo) of W1 (1024, 1440) and W2 (1024, 1024)
        self.hw = dict(w1=(_p(self.hw_planes[0]), _p(self.hw_planes[1])), w2=(_p(self.hw_planes[0]) + 2 * o2, _p(self.hw_planes[1]) + 2 * o2))
        return _abi.SplitItem(src=_p(w1), hi=_p(self.hw_planes[0]), lo=_p(self.hw_planes[1]), rows=1, cols=span, ld_src=span, ld_out=span, transpose=0)

    def _lean(self, pl, train, W):
        """the default single-process training plans (conv stack recomputed from the token rows + the head on plane GEMMs) touch NO accumulator that must be
        cleared beforehand: BatchNorm statistics travel as per-sample partial rows, K splits as slabs -- no arena memset, hence no second-stream work and no
        join at the start of the step"""
        return bool(train and W == 1 and self._cstack_enabled(pl) and self._head_planes_enabled(pl))

    def _head_gemm(self, pl, b, tag, a_planes, K, w, N, B, kmajor=False):
        """one K-parallel head GEMM: (B, K) planes x the weight planes `w` -- (N, K), k contiguous, or with kmajor (K, N), n contiguous -> partial slabs
        b[tag] (slices, B, N); returns (slab pointer, slice count, slab stride) for the launch that consumes them"""
        S = int(lib().eegclip_head_gemm_slices(B, N, K))
        if tag not in b or b[tag].shape != (S, B, N):
            b[tag] = torch.empty(S, B, N, dtype=torch.float32, device=self.device)
        pl.call_desc("eegclip_head_gemm", _abi.HeadGemmDesc(a_hi=a_planes[0], a_lo=a_planes[1], b_hi=w[0], b_lo=w[1], lda=K, ldb=N if kmajor else K, M=B, N=N, K=K,
                                                           slices=S, slab_stride=B * N, C=_p(b[tag]), ldc=N, b_kmajor=int(kmajor)))
        return _p(b[tag]), S, B * N

    def _token_block_enabled(self, pl):
        """the fused transformer-block forward (csrc/token_block.hip) in the default split-bf16 arithmetic -- since round 4 also for the joint-subject
        model (its value embedding is a per-sample weight base inside the kernel); EEGCLIP_TOKEN_BLOCK=0 pins the launch-per-Linear plan
        (diagnosis, A/B timing)"""
        if self.joint and self.n_subj > WGRAD_TOK_MAX_PROBLEMS:
            return False        # the per-subject value-embedding gradients are ONE eegclip_wgrad_tok launch (<= 12 problems): larger subject tables take the
            #                     grouped-GEMM plans, which split into per-member launches (the reference takes any num_subjects, Embed.py:127-131)
        return pl.precision == _abi.PREC_BF16X3 and os.environ.get("EEGCLIP_TOKEN_BLOCK", "1") != "0"

    def _token_planes(self, b, B, *names):
        """token-plane tensors (csrc/wgrad_tok.hip layout: per sample [hi | lo][64 tokens][256 channels] bf16) the fused kernels write for the
        weight-gradient GEMMs instead of fp32"""
        for n in names:
            if n not in b:
                b[n] = torch.empty((3, B, 2, L_TOK, 256) if n == "dqkvp" else (B, 2, L_TOK, 256), dtype=torch.bfloat16, device=self.device)
        return [_p(b[n]) for n in names]

    def saved_f32(self, B, name):
        """fp32 view of a saved activation whichever form the plans of this engine keep it in (tests, feature inspection): the fused transformer
        block leaves ctx / n1 / g1 / df2 / dg1 / da1 / dqkv as token planes (hi + lo = the value the split-bf16 GEMMs see)"""
        b = self.bufs[B]
        key = name + "p"
        if key not in b:
            return b[name]
        t = b[key].to(torch.float32)
        v = t[..., 0, :, :] + t[..., 1, :, :]                       # (..., B, 64, 256)
        if name in ("ctx", "dqkv"):                                  # channel 64 head + d -> column 62 head + d
            v = v.reshape(*v.shape[:-1], N_HEADS, 64)[..., :D_HEAD].reshape(*v.shape[:-1], HE)
            if name == "dqkv":
                v = v.permute(1, 2, 0, 3).reshape(B * L_TOK, 3 * HE)
                return v
        else:
            v = v[..., :D_FF if name in ("g1", "dg1") else D_MODEL]
        return v.reshape(B * L_TOK, -1)

    def stale(self, model):
        a, b = self._check
        return (a.data_ptr() != self.flat.data_ptr() or b.data.data_ptr() != self.P[_CHECK_KEY].data_ptr()
                or model.logit_scale.device != self.device)

    # ---- buffers -----------------------------------------------------------------------------------------------
    def _alloc(self, B):
        dev = self.device

        def f(*s):
            return torch.empty(*s, dtype=torch.float32, device=dev)

        R = B * L_TOK
        b = dict(
            h=f(B, L_TOK, D_MODEL), qkv=f(R, 3 * HE), ctx=f(R, HE), r1=f(R, D_MODEL), n1=f(R, D_MODEL), mu1=f(R), rs1=f(R),
            f1=f(R, D_FF), g1=f(R, D_FF), r2=f(R, D_MODEL), n2=f(R, D_MODEL), mu2=f(R), rs2=f(R), n3=f(B, L_TOK, D_MODEL), mu3=f(R), rs3=f(R),
            z2=f(B, C_TS, W_TS),
            feat=f(B, F_TS), u=f(B, P_DIM), gu=f(B, P_DIM), s=f(B, P_DIM), out=f(B, P_DIM), mu4=f(B), rs4=f(B),
            bn=f(4, C_TS), ids=torch.zeros(B, dtype=torch.long, device=dev),
        )
        if self.joint:         # subject-ordered copies for batches that arrive in another order (see _build_fwd)
            jmeta = torch.zeros(2 * B, dtype=torch.int32, device=dev)      # [subject of sample b | the batch's sample numbers ordered by subject]
            b.update(xs=f(B, N_CH, T_LEN), hs=f(B, L_TOK, D_MODEL), jmeta=jmeta, subj32=jmeta[:B], perm=jmeta[B:])
        # everything a plan must clear before use lives in two arenas (forward / backward): ONE memset each instead of five
        nsum = 2 * 2 * C_TS                                        # two BatchNorm sum rows of 2C doubles per direction
        ny2 = (B * C_TS * W_TS + 1) // 2                          # y2 (B,40,36) f32: the K-split spatial conv accumulates into it
        nzf, nzb = nsum + B * P_DIM + ny2, nsum + (B * P_DIM + B * F_TS + 1) // 2
        # the two arenas are one allocation: a TRAINING forward clears both with one memset (its backward follows), see backward()
        zfb = torch.zeros(nzf + nzb, dtype=torch.float64, device=dev)
        zf, zb = zfb[:nzf], zfb[nzf:]            # fwd: sums[0..1] | hacc (2,B,P_DIM) f32 | y2         bwd: sums[2..3] | dgu | dfeat
        sf, sb = zf[:nsum].view(2, 2 * C_TS), zb[:nsum].view(2, 2 * C_TS)
        zbf = zb[nsum:].view(torch.float32)
        b.update(zf=zf, zb=zb, zfb=zfb, zb_clean=True, sums=[sf[0], sf[1], sb[0], sb[1]], hacc=zf[nsum:nsum + B * P_DIM].view(torch.float32).view(2, B, P_DIM),
                 y2=zf[nsum + B * P_DIM:].view(torch.float32)[:B * C_TS * W_TS].view(B, C_TS, W_TS),
                 dgu=zbf[:B * P_DIM].view(B, P_DIM), dfeat=zbf[B * P_DIM:B * P_DIM + B * F_TS].view(B, F_TS))
        return b

    def _alloc_bwd(self, B, b):
        dev = self.device

        def f(*s):
            return torch.empty(*s, dtype=torch.float32, device=dev)

        R = B * L_TOK
        b.update(ds=f(B, P_DIM), dv=f(B, P_DIM), dz2=f(B, C_TS, W_TS), dy2=f(B, C_TS, W_TS),
                 # tsconv_bwd_x writes token rows 0..62 of every sample; row 63 (EEG channel 62, dropped by the reference's [:, :63]
                 # slice) never receives a gradient from the conv path: zeroed once here, nothing else ever writes dn3
                 dn3=torch.zeros(B, L_TOK, D_MODEL, dtype=torch.float32, device=dev),
                 dn2=f(R, D_MODEL), dr2=f(R, D_MODEL), df2=f(R, D_MODEL), dg1=f(R, D_FF), dr1=f(R, D_MODEL), da1=f(R, D_MODEL),
                 dctx=f(R, HE), dqkv=f(R, 3 * HE))

    # ---- forward plan -----------------------------------------------------------------------------------------
    def _build_fwd(self, B, train, shared, probs):
        P, b = self.P, self.bufs[B]
        pe_, pc_, pp_ = probs
        pl = Plan(f"atms_fwd[B={B}]")
        R = B * L_TOK
        pe = self.buffers[_E + "position_embedding.pe"]
        pl.tb_desc = None
        cstack = self._cstack_enabled(pl)
        head_planes = self._head_planes_enabled(pl)
        W = self._world() if train else 1          # data-parallel SyncBN: batch statistics over the GLOBAL batch
        lean = self._lean(pl, train, W)
        pl.lean = lean
        if cstack:
            # nothing before the conv stack needs it: the arena clear goes to the second stream, under the transformer block; the main stream joins in front of
            # the conv stack.  (The lean plans have nothing to clear.)
            if not lean:
                pl.memset(b["zfb"] if train else b["zf"], side=os.environ.get("EEGCLIP_START_SIDE", "1") != "0")
                pl.clears_zb = train
            if not hasattr(self, "cs_packed"):
                self.cs_packed = torch.empty(int(lib().eegclip_cstack_packed_bytes(N_CH)) // 2, dtype=torch.bfloat16, device=self.device)
                self.cs_packed_t = torch.empty(int(lib().eegclip_cstack_packed_t_bytes(N_CH)) // 2, dtype=torch.bfloat16, device=self.device)
            cs_prep = (_p(P[_TS + "4.weight"]), _p(self.cs_packed), _p(self.cs_packed_t), N_CH)       # (both sets: an eval-mode forward may be followed by a backward too)
            if not self._token_block_enabled(pl):
                pl.call("eegclip_cstack_pack_all", *cs_prep)
        if self._token_block_enabled(pl):
            # A1-A3 in ONE launch, one workgroup per sample (csrc/token_block.hip): the ten launches below it replace were bound by per-launch
            # prologue / epilogue and activation round trips, not by their K ~ 250 contractions
            if not hasattr(self, "tb_packed"):
                self.tb_packed = torch.empty(int(lib().eegclip_token_block_packed_bytes()) // 2, dtype=torch.bfloat16, device=self.device)
            ve_w0, ve_b0 = (self.ve_keys[0] if self.joint else (_E + "value_embedding.weight", _E + "value_embedding.bias"))
            tb_prep = (_p(P[ve_w0]), _p(P[_LY + "attention.query_projection.weight"]), _p(P[_LY + "attention.out_projection.weight"]),
                       _p(P[_LY + "conv1.weight"]), _p(P[_LY + "conv2.weight"]), _p(self.tb_packed))
            if cstack:      # the step's whole weight preparation in one launch (the fused block's packed matrices + the conv stack's fragments)
                pl.call("eegclip_weight_prep", *tb_prep, *cs_prep)
            else:
                pl.call("eegclip_token_block_pack", *tb_prep)
            joint_args = {}
            if self.joint:
                # joint-subject model (Embed.py:127-131,142-144): every subject's value embedding packed behind one another (they are equally spaced in
                # the flat parameter buffer); the kernel takes sample b's matrix and bias from its subject id -- no subject-ordered copy of the batch
                ve_stride = (P[self.ve_keys[1][0]].data_ptr() - P[ve_w0].data_ptr()) // 4 if self.n_subj > 1 else D_MODEL * T_LEN
                assert all(P[w].data_ptr() - P[ve_w0].data_ptr() == 4 * ve_stride * i and P[bk].data_ptr() - P[ve_b0].data_ptr() == 4 * ve_stride * i
                           for i, (w, bk) in enumerate(self.ve_keys)), "value embeddings are not equally spaced in the flat buffer"
                if not hasattr(self, "tb_packed_embed"):
                    self.tb_packed_embed = torch.empty(int(lib().eegclip_token_block_packed_embed_bytes(self.n_subj)) // 2, dtype=torch.bfloat16,
                                                       device=self.device)
                pl.call("eegclip_token_block_pack_embed", _p(P[ve_w0]), ve_stride, self.n_subj, _p(self.tb_packed_embed))
                joint_args = dict(packed_embed=_p(self.tb_packed_embed), embed_subject=_p(b["subj32"]), bv_stride=ve_stride)
            tok = P[_TOK_SHARED] if shared else P[_TOK_TABLE]
            # the X operands of the block's weight gradients leave as token planes (what the kernel holds in LDS), not fp32: h keeps its fp32 copy
            # too (the residual of the attention sublayer re-reads it), ctx / n1 / g1 exist only as planes; n2 is re-evaluated by the backward
            xp, hp, ctxp, n1p, g1p = self._token_planes(b, B, "xp", "hp", "ctxp", "n1p", "g1p")
            # (round 6) training plans: the conv stack's BatchNorm1 batch sums of a sample are the TAIL of the block kernel's workgroup (the n3 rows it has just
            # written are in L2; csrc/cstack_common.h: cs_stats1_sample) instead of the eegclip_cstack_stats1 launch.  EEGCLIP_STATS1_TAIL=0: the launch (A/B aid)
            stats_args = {}
            pl.stats1_in_block = bool(cstack and train and os.environ.get("EEGCLIP_STATS1_TAIL", "1") != "0")
            if pl.stats1_in_block:
                if "cs_rows" not in b:
                    b["cs_rows"] = torch.empty(2, B, 2 * C_TS, dtype=torch.float64, device=self.device)
                stats_args = dict(cs_w25=_p(P[_TS + "0.weight"]), cs_bias=_p(P[_TS + "0.bias"]), cs_rows=_p(b["cs_rows"][0]), cs_H=N_CH)
            pl.tb_desc = pl.call_desc("eegclip_token_block_fwd", _abi.TokenBlockDesc(
                B=B, x=0, packed=_p(self.tb_packed), bv=_p(P[ve_b0]), pe=_p(pe), tokens=_p(tok), ids=None if shared else _p(b["ids"]), **joint_args, **stats_args,
                bqkv=_p(P[_LY + "attention.query_projection.bias"]), bo=_p(P[_LY + "attention.out_projection.bias"]), ln1_g=_p(P[_LY + "norm1.weight"]),
                ln1_b=_p(P[_LY + "norm1.bias"]), b1=_p(P[_LY + "conv1.bias"]), b2=_p(P[_LY + "conv2.bias"]), ln2_g=_p(P[_LY + "norm2.weight"]),
                ln2_b=_p(P[_LY + "norm2.bias"]), ln3_g=_p(P["encoder.encoder.norm.weight"]), ln3_b=_p(P["encoder.encoder.norm.bias"]),
                h=_p(b["h"]), qkv=_p(b["qkv"]), r1=_p(b["r1"]), mu1=_p(b["mu1"]), rs1=_p(b["rs1"]), f1=_p(b["f1"]),
                r2=_p(b["r2"]), n2=None, mu2=_p(b["mu2"]), rs2=_p(b["rs2"]), n3=_p(b["n3"]), mu3=_p(b["mu3"]), rs3=_p(b["rs3"]),
                xp=xp, hp=hp, ctxp=ctxp, n1p=n1p, g1p=g1p,
                drop_p=pe_, eps=EPS, scale=1.0 / math.sqrt(D_HEAD), seed=0, site_embed=SITE_EMBED, site_attn=SITE_ATTN, site_attn_out=SITE_ATTN_OUT,
                site_ffn_act=SITE_FFN_ACT, site_ffn_out=SITE_FFN_OUT), seeded=pe_ > 0.0)
        else:
            # A1: value embedding + PE into token rows 1..63, then subject token + dropout      (Embed.py:146-162)
            hmap = D(D_MODEL, div=N_CH, so=L_TOK * D_MODEL)        # GEMM row m = (sample, channel) -> token row 1 + channel of that sample
            if not self.joint:
                pl.x_gemm = pl.gemm(B * N_CH, D_MODEL, T_LEN, 0, D(T_LEN), D(1), _p(P[_E + "value_embedding.weight"]), D(1), D(T_LEN),
                        _p(b["h"]) + 4 * D_MODEL, hmap, D(1), bias_n=_p(P[_E + "value_embedding.bias"]),
                        R=_p(pe), Rm=D(D_MODEL, div=N_CH, so=0), Rn=D(1))
            else:
                # joint-subject model (Embed.py:142-144): one GEMM per subject over that subject's block of the subject-ordered batch.  A batch
                # that is not already in subject order is gathered into xs first and the token rows are scattered back to batch order after
                # (ops skipped otherwise); M and the A / C pointers of each GEMM are patched per call (_joint_layout), absent subjects skipped.
                XR = N_CH * T_LEN
                pl.j_gather = len(pl.ops)
                pl.call("eegclip_gather_rows", _p(b["xs"]), XR, 0, XR, _p(b["perm"]), B, XR, 0)
                pl.j_gemm = [pl.desc(N_CH, D_MODEL, T_LEN, 0, D(T_LEN), D(1), _p(P[w]), D(1), D(T_LEN), 0, hmap, D(1), bias_n=_p(P[bk]),
                                     R=_p(pe), Rm=D(D_MODEL, div=N_CH, so=0), Rn=D(1)) for w, bk in self.ve_keys]
                pl.j_arr, pl.j_group = pl.gemm_grouped(self.n_subj)             # ONE launch for all subjects of the batch
                pl.j_scatter = len(pl.ops)
                pl.call("eegclip_gather_rows", _p(b["h"]) + 4 * D_MODEL, L_TOK * D_MODEL, _p(b["hs"]) + 4 * D_MODEL, L_TOK * D_MODEL, _p(b["perm"]), B,
                        N_CH * D_MODEL, 1)
            tok = P[_TOK_SHARED] if shared else P[_TOK_TABLE]
            pl.call("eegclip_embed_finish", _p(b["h"]), _p(tok), None if shared else _p(b["ids"]), B, L_TOK, D_MODEL, pe_, 0, SITE_EMBED, seed_at=7)
            # A2: fused QKV projection (weights adjacent in the flat buffer) + attention      (SelfAttention_Family.py:199-213)
            pl.gemm(R, 3 * HE, D_MODEL, _p(b["h"]), D(D_MODEL), D(1), _p(P[_LY + "attention.query_projection.weight"]), D(1), D(D_MODEL),
                    _p(b["qkv"]), D(3 * HE), D(1), bias_n=_p(P[_LY + "attention.query_projection.bias"]))
            pl.call("eegclip_attention_fwd", _p(b["qkv"]), _p(b["ctx"]), B, L_TOK, N_HEADS, D_HEAD, 3 * HE, 1.0 / math.sqrt(D_HEAD), pe_, 0,
                    SITE_ATTN, seed_at=9)
            # (dropout + residual of both sublayers live in the LayerNorm kernel that follows, not in the GEMM epilogue: one Philox block per 4
            #  consecutive columns there, one per ELEMENT in an MFMA accumulator layout -- 12 us per GEMM)
            pl.gemm(R, D_MODEL, HE, _p(b["ctx"]), D(HE), D(1), _p(P[_LY + "attention.out_projection.weight"]), D(1), D(HE),
                    _p(b["r1"]), D(D_MODEL), D(1), bias_n=_p(P[_LY + "attention.out_projection.bias"]))
            # A3: post-LN encoder layer + final LN      (Transformer_EncDec.py:45-51,77-78)
            pl.call("eegclip_residual_layernorm_fwd", _p(b["r1"]), _p(b["h"]), _p(b["r1"]), pe_, 0, SITE_ATTN_OUT, _p(P[_LY + "norm1.weight"]),
                    _p(P[_LY + "norm1.bias"]), _p(b["n1"]), _p(b["mu1"]), _p(b["rs1"]), None, None, None, None, None, R, D_MODEL, EPS, seed_at=4)
            pl.gemm(R, D_FF, D_MODEL, _p(b["n1"]), D(D_MODEL), D(1), _p(P[_LY + "conv1.weight"]), D(1), D(D_MODEL), _p(b["g1"]), D(D_FF), D(1),
                    Cpre=_p(b["f1"]), bias_n=_p(P[_LY + "conv1.bias"]), act=ACT_GELU, drop_p=pe_, drop_site=SITE_FFN_ACT)
            pl.gemm(R, D_MODEL, D_FF, _p(b["g1"]), D(D_FF), D(1), _p(P[_LY + "conv2.weight"]), D(1), D(D_FF), _p(b["r2"]), D(D_MODEL), D(1),
                    bias_n=_p(P[_LY + "conv2.bias"]))
            # norm2 and the encoder's final norm back to back in one launch
            pl.call("eegclip_residual_layernorm_fwd", _p(b["r2"]), _p(b["n1"]), _p(b["r2"]), pe_, 0, SITE_FFN_OUT, _p(P[_LY + "norm2.weight"]),
                    _p(P[_LY + "norm2.bias"]), _p(b["n2"]), _p(b["mu2"]), _p(b["rs2"]), _p(P["encoder.encoder.norm.weight"]),
                    _p(P["encoder.encoder.norm.bias"]), _p(b["n3"]), _p(b["mu3"]), _p(b["rs3"]), R, D_MODEL, EPS, seed_at=4)
        # A4+A5: tokens 0..62 -> box filter + 25-tap stride-5 conv (= conv + avg-pool) -> BN -> ELU      (ATMS_retrieval.py:91,102-105)
        sums, bn = b["sums"], b["bn"]
        bn2_rows = None
        if cstack:
            if not lean:
                pl.join()
            bn2_rows = self._build_fwd_cstack(pl, b, B, train, W)
        else:
            pl.memset(b["zfb"] if train else b["zf"])
            pl.clears_zb = train
            self._build_fwd_conv_y1(pl, b, B, train, W)
        # BN2 -> ELU -> dropout -> 1x1 conv + 'b e h w -> b (h w) e' + flatten: feat[b, w*40+e], one workgroup per sample      (:107-114,145)
        run2 = (_p(self.buffers[_TS + "5.running_mean"]), _p(self.buffers[_TS + "5.running_var"]), _p(self.buffers[_TS + "5.num_batches_tracked"]))
        if head_planes:
            # (round 6) ... and feat again as bf16 hi | lo planes: the A operand of the head's first Linear.  With per-sample BatchNorm2 partial rows the
            # finalize rides in this kernel's prologue; otherwise mean / rstd are inputs
            if "featp" not in b:
                # (+ 128 elements of slack: eegclip_wgrad_planes reads whole 128-channel tiles -- up to channel 1535 of the LAST row of the lo plane)
                b["featp_buf"] = torch.zeros(2 * B * F_TS + 128, dtype=torch.bfloat16, device=self.device)
                b["featp"] = b["featp_buf"][:2 * B * F_TS].view(2, B, F_TS)
                b["gup"] = torch.empty(2, B, P_DIM, dtype=torch.bfloat16, device=self.device)
            rows_args = (_p(bn2_rows), B, float(B * W_TS), EPS, 0.1, _p(bn[2]), _p(bn[3]), *run2) if bn2_rows is not None else \
                (None, 0, 1.0, EPS, 0.1, _p(bn[2]), _p(bn[3]), None, None, None)
            # riders (arguments 23 / 24): the head weights' plane split by extra workgroups of this launch; the step plan adds the loss targets
            pl.rider_items = (_abi.SplitItem * 4)()
            pl.rider_items[0] = self._head_weight_planes()
            pl._keep.append(pl.rider_items)
            pl.rider_op = len(pl.ops)
            pl.call("eegclip_proj1x1_fwd_rows_planes", _p(b["y2"]), *rows_args, _p(P[_TS + "5.weight"]), _p(P[_TS + "5.bias"]),
                    _p(P["enc_eeg.0.projection.0.weight"]), _p(P["enc_eeg.0.projection.0.bias"]), _p(b["z2"]), _p(b["feat"]), B, pc_, 0, SITE_CONV,
                    _p(b["featp"][0]), _p(b["featp"][1]), pl.rider_items, 1, seed_at=19)
        elif bn2_rows is not None:
            # (the BatchNorm2 finalize rides in this kernel's prologue: the batch statistics as the per-sample partial rows eegclip_cstack_fwd left)
            pl.call("eegclip_proj1x1_fwd_rows", _p(b["y2"]), _p(bn2_rows), B, float(B * W_TS), EPS, 0.1, _p(bn[2]), _p(bn[3]), *run2,
                    _p(P[_TS + "5.weight"]), _p(P[_TS + "5.bias"]), _p(P["enc_eeg.0.projection.0.weight"]), _p(P["enc_eeg.0.projection.0.bias"]),
                    _p(b["z2"]), _p(b["feat"]), B, pc_, 0, SITE_CONV, seed_at=19)
        else:
            pl.call("eegclip_proj1x1_fwd", _p(b["y2"]), _p(bn[2]), _p(bn[3]), _p(P[_TS + "5.weight"]), _p(P[_TS + "5.bias"]),
                    _p(P["enc_eeg.0.projection.0.weight"]), _p(P["enc_eeg.0.projection.0.bias"]), _p(b["z2"]), _p(b["feat"]), B, pc_, 0, SITE_CONV,
                    seed_at=11)
        pl.head_planes = head_planes
        if head_planes:
            # A6: projection head      (:157-167) on the K-parallel plane GEMM (csrc/head_gemm.hip, round 6): M = B is small, so K is split over workgroups --
            # each slice leaves its partial tile as a slab and the launch that consumes the result anyway adds the slabs while it loads them:
            # bias + GELU (+ the planes of gelu(u) for the second Linear) after the first, dropout + residual + LayerNorm after the second
            fp, gp = (_p(b["featp"][0]), _p(b["featp"][1])), (_p(b["gup"][0]), _p(b["gup"][1]))
            # (OPT-IN EEGCLIP_HEAD_FUSED_EPI=1: the first Linear unsplit with bias + GELU + planes in its own epilogue -- one launch instead of GEMM slabs +
            #  head_act; measured SLOWER in the step, 0.658 - 0.661 against 0.647 - 0.655 ms alternated in one call (34 launches against 36): the unsplit GEMM walks
            #  K alone for 13 us where the K-parallel GEMM + the elementwise launch take 6.5 + 6)
            fused_epi = os.environ.get("EEGCLIP_HEAD_FUSED_EPI", "0") == "1"
            if fused_epi:
                pl.call_desc("eegclip_head_gemm", _abi.HeadGemmDesc(a_hi=fp[0], a_lo=fp[1], b_hi=self.hw["w1"][0], b_lo=self.hw["w1"][1], lda=F_TS, ldb=F_TS, M=B,
                                                                   N=P_DIM, K=F_TS, slices=1, slab_stride=0, bias=_p(P["proj_eeg.0.bias"]), Cpre=_p(b["u"]),
                                                                   ldcpre=P_DIM, act=ACT_GELU, C=_p(b["gu"]), ldc=P_DIM, p_hi=gp[0], p_lo=gp[1], ldp=P_DIM))
            else:
                sl1 = self._head_gemm(pl, b, "hslab1", fp, F_TS, self.hw["w1"], P_DIM, B)
                pl.call("eegclip_head_act", *sl1, _p(P["proj_eeg.0.bias"]), _p(b["u"]), _p(b["gu"]), *gp, B, P_DIM)
            sl2 = self._head_gemm(pl, b, "hslab2", gp, P_DIM, self.hw["w2"], P_DIM, B)
            # s = u + dropout(W gelu(u) + b), out = LayerNorm(s): ResidualAdd + LayerNorm of Proj_eeg in one launch; `out` is a fresh tensor per call
            # (argument 8 is patched by forward()); arguments 19 / 20: out again as planes (the step plan's loss operand)
            pl.out_op = len(pl.ops)
            pl.call("eegclip_residual_layernorm_fwd_slabs", sl2[0], _p(b["u"]), _p(b["s"]), pp_, 0, SITE_PROJ, _p(P["proj_eeg.2.weight"]),
                    _p(P["proj_eeg.2.bias"]), 0, _p(b["mu4"]), _p(b["rs4"]), None, None, None, None, None, B, P_DIM, EPS, None, None, sl2[1], sl2[2],
                    _p(P["proj_eeg.1.fn.1.bias"]), seed_at=4)
            return pl
        # A6 on fp32 operands (exact-fp32 plans, EEGCLIP_HEAD_GEMM=0): a 4 x 16 tile grid cannot fill 256 CUs and each workgroup walks K = 1440
        # serially, so the products are split over K (atomics into a zeroed buffer) and bias/GELU/dropout/residual run as a tiny epilogue.
        skh = _head_split(B)
        if skh > 1:
            pl.gemm(B, P_DIM, F_TS, _p(b["feat"]), D(F_TS), D(1), _p(P["proj_eeg.0.weight"]), D(1), D(F_TS), _p(b["hacc"][0]), D(P_DIM), D(1),
                    accumulate=1, split_k=skh)
            pl.call("eegclip_bias_act", _p(b["hacc"][0]), _p(P["proj_eeg.0.bias"]), _p(b["u"]), None, _p(b["gu"]), B, P_DIM, ACT_GELU, 0.0, 0, 0)
            pl.gemm(B, P_DIM, P_DIM, _p(b["gu"]), D(P_DIM), D(1), _p(P["proj_eeg.1.fn.1.weight"]), D(1), D(P_DIM), _p(b["hacc"][1]), D(P_DIM), D(1),
                    bias_n=_p(P["proj_eeg.1.fn.1.bias"]), accumulate=1, split_k=skh)             # (slice 0 adds the bias)
            w_lin = b["hacc"][1]
        else:
            pl.gemm(B, P_DIM, F_TS, _p(b["feat"]), D(F_TS), D(1), _p(P["proj_eeg.0.weight"]), D(1), D(F_TS), _p(b["gu"]), D(P_DIM), D(1),
                    Cpre=_p(b["u"]), bias_n=_p(P["proj_eeg.0.bias"]), act=ACT_GELU)
            pl.gemm(B, P_DIM, P_DIM, _p(b["gu"]), D(P_DIM), D(1), _p(P["proj_eeg.1.fn.1.weight"]), D(1), D(P_DIM), _p(b["s"]), D(P_DIM), D(1),
                    bias_n=_p(P["proj_eeg.1.fn.1.bias"]))
            w_lin = b["s"]
        # s = u + dropout(W gelu(u) + b), out = LayerNorm(s): ResidualAdd + LayerNorm of Proj_eeg in one launch; `out` is a fresh tensor per
        # call (argument 8 is patched by forward()), so callers keep what they are handed and no copy is made
        pl.out_op = len(pl.ops)
        pl.call("eegclip_residual_layernorm_fwd", _p(w_lin), _p(b["u"]), _p(b["s"]), pp_, 0, SITE_PROJ, _p(P["proj_eeg.2.weight"]),
                _p(P["proj_eeg.2.bias"]), 0, _p(b["mu4"]), _p(b["rs4"]), None, None, None, None, None, B, P_DIM, EPS, seed_at=4)
        return pl

    def _cstack_enabled(self, pl):
        """the conv stack recomputed from the token rows (csrc/cstack*.hip: y1 never written) in split-bf16 plans; exact-fp32 plans
        (EEGCLIP_GEMM_PRECISION=f32) keep the round-4 kernels around y1 in HBM"""
        from .plan import default_gemm_precision
        precision = pl.precision if pl is not None else default_gemm_precision()
        return precision == _abi.PREC_BF16X3

    def _build_fwd_cstack(self, pl, b, B, train, W):
        """A4+A5 on csrc/cstack.hip (round 5): BatchNorm1 sums WITHOUT writing y1, then per sample y1 tile -> BN1 -> ELU -> spatial conv chained on the
        bf16 matrix cores; the BatchNorm finalizes ride in the consumers' prologues / one rows kernel.      (ATMS_retrieval.py:91,102-106)"""
        P, sums, bn = self.P, b["sums"], b["bn"]
        dev = self.device
        if "cs_rows" not in b:
            b["cs_rows"] = torch.empty(2, B, 2 * C_TS, dtype=torch.float64, device=dev)          # BatchNorm1 | BatchNorm2 partial rows, one per sample
        rows1, rows2 = b["cs_rows"][0], b["cs_rows"][1]
        count1 = float(W * B * N_CH * W_TS)
        stat1, nstat1 = None, 0
        if train:
            if not getattr(pl, "stats1_in_block", False):        # (else: row b of rows1 was written by sample b's workgroup of the transformer-block kernel)
                pl.call("eegclip_cstack_stats1", _p(b["n3"]), L_TOK * D_MODEL, D_MODEL, _p(P[_TS + "0.weight"]), _p(P[_TS + "0.bias"]), _p(rows1), B, N_CH)
            stat1, nstat1 = _p(rows1), B
            if W > 1:
                pl.call("eegclip_bn_finalize_rows", _p(rows1), B, count1, EPS, 0.1, C_TS, None, None, None, None, None, _p(sums[0]))
                pl.callback(lambda: self._allreduce(sums[0]), "allreduce_bn1")
                stat1, nstat1 = _p(sums[0]), 1
        else:
            pl.call("eegclip_bn_finalize", _p(sums[0]), count1, EPS, 0.1, C_TS, _p(bn[0]), _p(bn[1]),
                    _p(self.buffers[_TS + "2.running_mean"]), _p(self.buffers[_TS + "2.running_var"]), 0,
                    _p(self.buffers[_TS + "2.num_batches_tracked"]))
        pl.call_desc("eegclip_cstack_fwd", _abi.CstackFwdDesc(
            B=B, H=N_CH, x=_p(b["n3"]), xs_b=L_TOK * D_MODEL, xs_h=D_MODEL, w25=_p(P[_TS + "0.weight"]), bias1=_p(P[_TS + "0.bias"]), stat1=stat1,
            nstat1=nstat1, count1=count1, eps=EPS, momentum=0.1, gamma1=_p(P[_TS + "2.weight"]), beta1=_p(P[_TS + "2.bias"]), mean1=_p(bn[0]),
            rstd1=_p(bn[1]), run_mean1=_p(self.buffers[_TS + "2.running_mean"]) if train else None,
            run_var1=_p(self.buffers[_TS + "2.running_var"]) if train else None,
            nbt1=_p(self.buffers[_TS + "2.num_batches_tracked"]) if train else None, packed=_p(self.cs_packed), bias2=_p(P[_TS + "4.bias"]),
            y2=_p(b["y2"]), stat2=_p(rows2) if train else None))
        count2 = float(W * B * W_TS)
        run2 = (_p(self.buffers[_TS + "5.running_mean"]), _p(self.buffers[_TS + "5.running_var"]))
        nbt2 = _p(self.buffers[_TS + "5.num_batches_tracked"])
        if train and W == 1:
            return rows2                            # finalized in the prologue of eegclip_proj1x1_fwd_rows
        else:
            if train:
                pl.call("eegclip_bn_finalize_rows", _p(rows2), B, count2, EPS, 0.1, C_TS, None, None, None, None, None, _p(sums[1]))
                pl.callback(lambda: self._allreduce(sums[1]), "allreduce_bn2")
            pl.call("eegclip_bn_finalize", _p(sums[1]), count2, EPS, 0.1, C_TS, _p(bn[2]), _p(bn[3]), *run2, int(train), nbt2)
        return None

    def _cstack_bwd_enabled(self, pl):
        """the conv-stack backward is recomputed from the token rows whenever the forward is (csrc/cstack_bwd.hip): y1 does not exist for anything else"""
        return self._cstack_enabled(pl)

    def _build_fwd_conv_y1(self, pl, b, B, train, W):
        """A4+A5 around y1 in HBM (rounds 1-4 kernels; exact-fp32 plans only)"""
        P, sums, bn = self.P, b["sums"], b["bn"]
        if "y1" not in b:              # the conv + pool output (B,40,63,36): exists in HBM only for the kernels around it
            b["y1"] = torch.empty(B, C_TS, N_CH, W_TS, dtype=torch.float32, device=self.device)
        if "tsf_ws" not in b:
            b["tsf_ws"] = torch.empty(int(lib().eegclip_tsconv_fwd_workspace_floats(B, N_CH)) // 2, dtype=torch.float64, device=self.device)
        pl.call("eegclip_tsconv_fwd", _p(b["n3"]), L_TOK * D_MODEL, D_MODEL, _p(P[_TS + "0.weight"]), _p(P[_TS + "0.bias"]), _p(b["y1"]), B, N_CH, T_LEN,
                C_TS, _p(sums[0]) if train else None, _p(b["tsf_ws"]))
        if W > 1:
            pl.callback(lambda: self._allreduce(sums[0]), "allreduce_bn1")
        pl.call("eegclip_bn_finalize", _p(sums[0]), float(W * B * N_CH * W_TS), EPS, 0.1, C_TS, _p(bn[0]), _p(bn[1]),
                _p(self.buffers[_TS + "2.running_mean"]), _p(self.buffers[_TS + "2.running_var"]), int(train),
                _p(self.buffers[_TS + "2.num_batches_tracked"]))
        if "scf_ws" not in b:          # the K-slice partial tiles of sconv_fwd: slabs summed by the statistics kernel of the same call (no atomics)
            b["scf_ws"] = torch.empty(int(lib().eegclip_sconv_fwd_workspace_floats(B)), dtype=torch.float32, device=self.device)
        # BN1 -> ELU -> spatial (63x1) conv in ONE kernel: z1 = ELU(BN(y1)) is re-evaluated while y1 is staged, never stored; the
        # BatchNorm2 batch sums of y2 are accumulated by the same kernel      (:104-107)
        pl.call("eegclip_sconv_fwd", _p(b["y1"]), _p(bn[0]), _p(bn[1]), _p(P[_TS + "2.weight"]), _p(P[_TS + "2.bias"]), _p(P[_TS + "4.weight"]),
                _p(P[_TS + "4.bias"]), _p(b["y2"]), _p(sums[1]) if train else None, B, N_CH, 1, _p(b["scf_ws"]))
        if W > 1:
            pl.callback(lambda: self._allreduce(sums[1]), "allreduce_bn2")
        pl.call("eegclip_bn_finalize", _p(sums[1]), float(W * B * W_TS), EPS, 0.1, C_TS, _p(bn[2]), _p(bn[3]),
                _p(self.buffers[_TS + "5.running_mean"]), _p(self.buffers[_TS + "5.running_var"]), int(train),
                _p(self.buffers[_TS + "5.num_batches_tracked"]))

    # ---- backward plan -----------------------------------------------------------------------------------------
    def _build_bwd(self, B, shared, probs, want_dx, early_reduce=False, train=True):
        """train=False: backward of an eval-mode forward (frozen-BatchNorm fine-tuning, saliency).  BatchNorm then normalises with the
        running statistics, which do not depend on the batch: dx = gamma*rstd*da with no batch-mean terms, and the conv biases in front of
        the two BatchNorms get real gradients (in train mode the mean subtraction cancels them exactly)."""
        P, G, b = self.P, self.G, self.bufs[B]
        pe_, pc_, pp_ = probs
        pl = Plan(f"atms_bwd[B={B}]")
        R = B * L_TOK
        sums, bn = b["sums"], b["bn"]
        # the small parameter-gradient reductions (conv taps, the block's LayerNorm rows) wait for the end of the backward and go to the second stream behind
        # ONE fork with the token-row gradient: a fork costs the main stream 4 - 13 us of idle queue (tools/step_timeline.py), as much as each of them runs
        defer_small = os.environ.get("EEGCLIP_DEFER_SMALL", "1") != "0"
        pl.deferred = []
        wsk = int(os.environ.get("EEGCLIP_WGRAD_SK", "0"))                         # tuning aid: K-slice count of the long-K weight gradients
        # attention backward: split-bf16 products (csrc/attention_x3.hip) in plans whose GEMM precision is bf16x3, the exact-fp32 MFMA kernel otherwise
        attn_bwd = "eegclip_attention_bwd_x3" if pl.precision == _abi.PREC_BF16X3 else "eegclip_attention_bwd"
        sk = lambda k: (wsk if (wsk > 0 and k >= 4096) else max(1, min(64, k // 512)))      # split-K for the reduce-over-batch weight-gradient GEMMs

        head_side = os.environ.get("EEGCLIP_HEAD_WGRAD_SIDE", "1") != "0"      # (A/B aid)

        def wgrad(name, dY, ldy, X, ldx, Nout, Nin, K, bias=None, side=head_side):
            """G[name] (Nout, Nin) += dY^T X   with dY (K, ldy), X (K, ldx) row-major; bias: G[bias] (Nout) += column sums of dY, taken
            from the A tiles the same launch stages (rowsum_a) instead of a separate pass over dY"""
            return pl.gemm(Nout, Nin, K, dY, D(1), D(ldy), X, D(ldx), D(1), _p(G[name]), D(Nin), D(1), accumulate=1, split_k=sk(K),
                           rowsum_a=_p(G[bias]) if bias else None, side=side)    # nobody reads a weight gradient before the optimizer

        ln_side = os.environ.get("EEGCLIP_LN_SIDE", "1") != "0"           # LayerNorm parameter-gradient kernels on the second stream (A/B aid)
        # the three token-block LayerNorms reduce their gamma / beta gradients through per-workgroup partial rows (one workspace: the three launches
        # are ordered on one stream) instead of 256-way contended atomics
        if "lnp_ws" not in b:
            b["lnp_ws"] = torch.empty(int(lib().eegclip_layernorm_bwd_params_workspace_floats(R, D_MODEL)), dtype=torch.float32, device=self.device)
        lnp_ws = _p(b["lnp_ws"])
        # ... or, fused (default): the input-gradient kernel of those LayerNorms leaves the partial rows itself -- one pass over dy / x instead of two
        if "lnf_ws" not in b:
            b["lnf_ws"] = torch.empty(int(lib().eegclip_layernorm_bwd_full_workspace_floats(R, D_MODEL)), dtype=torch.float32, device=self.device)
        lnf_ws = _p(b["lnf_ws"])
        # (the BatchNorm backward sums and the split-K accumulators dgu / dfeat -- arena zb -- were cleared by the training forward's memset)
        # head LayerNorm
        pl.dout_op = len(pl.ops)
        head_planes = self._head_planes_enabled(pl) and hasattr(self, "hw")
        pl.head_planes = head_planes
        dfeat_slabs = None
        if head_planes:
            # (round 6) the head's backward on the K-parallel plane GEMM.  The upstream gradient may itself be the partial slabs of the loss's query-gradient
            # GEMM (arguments 1 / 2, patched by the step plan); the LayerNorm backward leaves ds, dv = ds * mask / (1 - p) (fp32: the weight gradients'
            # operands), dv again as planes (the next GEMM's A operand) and the summed upstream gradient for its parameter half
            if "dvp" not in b:
                b["dvp"] = torch.empty(2, B, P_DIM, dtype=torch.bfloat16, device=self.device)
                b["dup"] = torch.empty(2, B, P_DIM, dtype=torch.bfloat16, device=self.device)
                b["dout_sum"] = torch.empty(B, P_DIM, dtype=torch.float32, device=self.device)
            vp, up = (_p(b["dvp"][0]), _p(b["dvp"][1])), (_p(b["dup"][0]), _p(b["dup"][1]))
            pl.call("eegclip_layernorm_bwd_slabs", 0, 1, 0, _p(b["s"]), _p(P["proj_eeg.2.weight"]), _p(b["mu4"]), _p(b["rs4"]), _p(b["ds"]), B, P_DIM,
                    _p(b["dv"]), *vp, None, pp_, 0, SITE_PROJ, seed_at=15)
            # the head's second-stream launches (LayerNorm parameter half, the two weight gradients: operands dv | gu and du | feat stay untouched until the
            # optimizer) are emitted TOGETHER in front of the conv stack's backward, next to its own second-stream launch: ONE fork instead of three -- every
            # fork idles the main queue for ~7 us (tools/step_timeline.py).  EEGCLIP_HEAD_FORKS=3: each as soon as its operands exist (A/B aid)
            one_fork = os.environ.get("EEGCLIP_HEAD_FORKS", "1") == "1"
            head_side_ops = []

            def later(fn):
                if one_fork:
                    head_side_ops.append(fn)
                else:
                    fn()

            def ln_params():
                pl.dout_par_op = len(pl.ops)
                pl.call("eegclip_layernorm_bwd", 0, _p(b["s"]), None, _p(b["mu4"]), _p(b["rs4"]), None,
                        _p(G["proj_eeg.2.weight"]), _p(G["proj_eeg.2.bias"]), B, P_DIM, 0, None, 0.0, 0, 0, side=ln_side)
            later(ln_params)
            # both weight gradients of the head from the planes that exist anyway (dv | gelu(u), du | feat: every operand row-major, contraction over the batch
            # rows) in ONE eegclip_wgrad_planes launch -- the fp32-operand GEMMs (k-strided operands) took 30 + 37 us alone and far longer beside the backward
            wplanes = B % 32 == 0 and os.environ.get("EEGCLIP_HEAD_WGRAD_PLANES", "1") != "0"
            if not wplanes:
                later(lambda: wgrad("proj_eeg.1.fn.1.weight", _p(b["dv"]), P_DIM, _p(b["gu"]), P_DIM, P_DIM, P_DIM, B, bias="proj_eeg.1.fn.1.bias"))
            if os.environ.get("EEGCLIP_HEAD_FUSED_EPI", "0") == "1":     # (opt-in, see the forward: GELU' + the residual in the unsplit GEMM's epilogue)
                pl.call_desc("eegclip_head_gemm", _abi.HeadGemmDesc(a_hi=vp[0], a_lo=vp[1], b_hi=self.hw["w2"][0], b_lo=self.hw["w2"][1], lda=P_DIM, ldb=P_DIM, M=B,
                                                                   N=P_DIM, K=P_DIM, slices=1, slab_stride=0, act=ACT_GELU_GRAD, aux=_p(b["u"]), ldaux=P_DIM,
                                                                   R=_p(b["ds"]), ldr=P_DIM, C=_p(b["ds"]), ldc=P_DIM, p_hi=up[0], p_lo=up[1], ldp=P_DIM, b_kmajor=1))
            else:
                sl = self._head_gemm(pl, b, "dgu_slabs", vp, P_DIM, self.hw["w2"], P_DIM, B, kmajor=True)
                pl.call("eegclip_head_act_bwd", *sl, _p(b["u"]), _p(b["ds"]), _p(b["ds"]), *up, B * P_DIM)          # ds := du = ds + dgu * gelu'(u)
            if wplanes:
                gp_, fp_ = (_p(b["gup"][0]), _p(b["gup"][1])), (_p(b["featp"][0]), _p(b["featp"][1]))
                probs = (_abi.WgradPlanesProblem * 2)(
                    _abi.WgradPlanesProblem(a_hi=vp[0], a_lo=vp[1], lda=P_DIM, b_hi=gp_[0], b_lo=gp_[1], ldb=P_DIM, rows=B, M=P_DIM, N=P_DIM,
                                            out=_p(G["proj_eeg.1.fn.1.weight"]), ldo=P_DIM, bias_out=_p(G["proj_eeg.1.fn.1.bias"]), slices=1),
                    _abi.WgradPlanesProblem(a_hi=up[0], a_lo=up[1], lda=P_DIM, b_hi=fp_[0], b_lo=fp_[1], ldb=F_TS, rows=B, M=P_DIM, N=F_TS,
                                            out=_p(G["proj_eeg.0.weight"]), ldo=F_TS, bias_out=_p(G["proj_eeg.0.bias"]), slices=1))
                pl._keep.append(probs)
                later(lambda: pl.call("eegclip_wgrad_planes", probs, 2, side=head_side))
            else:
                later(lambda: wgrad("proj_eeg.0.weight", _p(b["ds"]), P_DIM, _p(b["feat"]), F_TS, P_DIM, F_TS, B, bias="proj_eeg.0.bias"))
            dfeat_slabs = self._head_gemm(pl, b, "dfeat_slabs", up, P_DIM, self.hw["w1"], F_TS, B, kmajor=True)
        else:
            head_side_ops = []
            self._build_bwd_head_f32(pl, b, B, pp_, ln_side, wgrad)
        # 1x1 conv backward + BatchNorm2 / ELU / dropout backward statistics in one launch (dW, dbias, dz2, sums[2]); then -- after the SyncBN
        # all-reduce of the sums under torch.distributed -- the apply pass.  dgamma / dbeta take this rank's LOCAL sums, so the later
        # mean-all-reduce of the flat gradient (every rank's gradient is W x its share) reproduces the single-process value.
        W = self._world() if train else 1
        zsum = None
        if not train:
            # eval-mode BatchNorm backward through the SAME apply kernels: they form dx = gamma*rstd*(da - S0/count - xhat*S1/count) from the
            # `sums` argument and dgamma / dbeta from `sums_local`; a zero `sums` drops the batch-mean terms (= the running-statistics formula)
            zsum = torch.zeros(2 * C_TS, dtype=torch.float64, device=self.device)
            pl._keep.append(zsum)

            def conv_bias_grad(bias_key, gamma_key, rstd, s):
                # d bias[c] = sum over (b,h,w) of the gradient entering the BatchNorm input = gamma[c]*rstd[c]*sum(da)[c]   (40 values)
                return lambda: G[bias_key].add_((P[gamma_key] * rstd * s[:C_TS].to(torch.float32)))
        if "pj_ws" not in b:
            b["pj_ws"] = torch.empty(int(lib().eegclip_proj1x1_bwd_workspace_floats(B)) // 2, dtype=torch.float64, device=self.device)
        lean = dfeat_slabs is not None and self._lean(pl, train, W)
        pl.lean = lean
        if lean:
            # (round 6) the 1x1-conv backward leaves one partial row per sample; the BatchNorm2-backward apply pass adds the rows' 80 sums itself (fixed order,
            # every workgroup the same value): no cleared accumulator, and the reduction of the rows into dW / dbias -- read by the optimizer only -- leaves
            # the dX chain for the second stream
            if "pj_bn_rows" not in b:
                b["pj_bn_rows"] = torch.empty(B, 2 * C_TS, dtype=torch.float64, device=self.device)
            pl.call("eegclip_proj1x1_bwd_rows", *dfeat_slabs, _p(b["z2"]), _p(P["enc_eeg.0.projection.0.weight"]), _p(b["y2"]), _p(bn[2]), _p(bn[3]),
                    _p(P[_TS + "5.weight"]), _p(P[_TS + "5.bias"]), _p(b["dz2"]), _p(b["pj_ws"]), _p(b["pj_bn_rows"]), B, pc_, 0, SITE_CONV, seed_at=15)
            pl.call("eegclip_bn_elu_bwd_apply_rows", _p(b["dz2"]), _p(b["y2"]), _p(bn[2]), _p(bn[3]), _p(P[_TS + "5.weight"]), _p(P[_TS + "5.bias"]),
                    _p(b["pj_bn_rows"]), B, 2 * C_TS, 0, float(B * W_TS),
                    _p(b["dy2"]), _p(G[_TS + "5.weight"]), _p(G[_TS + "5.bias"]), B, C_TS, W_TS, pc_, 0, SITE_CONV, seed_at=18)
            for f in head_side_ops:
                f()
            pl.call("eegclip_proj1x1_bwd_reduce", _p(b["pj_ws"]), B, _p(G["enc_eeg.0.projection.0.weight"]), _p(G["enc_eeg.0.projection.0.bias"]), None, side=True)
        else:
            if dfeat_slabs is not None:
                pl.call("eegclip_proj1x1_bwd_slabs", *dfeat_slabs, _p(b["z2"]), _p(P["enc_eeg.0.projection.0.weight"]), _p(b["y2"]), _p(bn[2]), _p(bn[3]),
                        _p(P[_TS + "5.weight"]), _p(P[_TS + "5.bias"]), _p(b["dz2"]), _p(G["enc_eeg.0.projection.0.weight"]),
                        _p(G["enc_eeg.0.projection.0.bias"]), _p(sums[2]), _p(b["pj_ws"]), B, pc_, 0, SITE_CONV, seed_at=17)
            else:
                pl.call("eegclip_proj1x1_bwd", _p(b["dfeat"]), _p(b["z2"]), _p(P["enc_eeg.0.projection.0.weight"]), _p(b["y2"]), _p(bn[2]), _p(bn[3]),
                        _p(P[_TS + "5.weight"]), _p(P[_TS + "5.bias"]), _p(b["dz2"]), _p(G["enc_eeg.0.projection.0.weight"]),
                        _p(G["enc_eeg.0.projection.0.bias"]), _p(sums[2]), _p(b["pj_ws"]), B, pc_, 0, SITE_CONV, seed_at=15)
            local2 = None
            if W > 1:
                local2 = torch.zeros_like(sums[2])
                pl._keep.append(local2)

                def exchange2():
                    local2.copy_(sums[2])
                    self._allreduce(sums[2])
                pl.callback(exchange2, "allreduce_bn2_bwd")
            if not train:
                pl.callback(conv_bias_grad(_TS + "4.bias", _TS + "5.weight", bn[3], sums[2]), "conv2_bias_grad_eval")
            pl.call("eegclip_bn_elu_bwd_apply", _p(b["dz2"]), _p(b["y2"]), _p(bn[2]), _p(bn[3]), _p(P[_TS + "5.weight"]), _p(P[_TS + "5.bias"]),
                    _p(sums[2]) if train else _p(zsum), (_p(local2) if local2 is not None else None) if train else _p(sums[2]), float(W * B * W_TS), _p(b["dy2"]), _p(G[_TS + "5.weight"]), _p(G[_TS + "5.bias"]), B, C_TS,
                    W_TS, pc_, 0, SITE_CONV, seed_at=16)
            for f in head_side_ops:
                f()
        if self._cstack_bwd_enabled(pl):
            self._build_bwd_cstack(pl, b, B, train, W, zsum, conv_bias_grad if not train else None, early_reduce, defer_small)
        else:
            self._build_bwd_conv_y1(pl, b, B, train, W, zsum, conv_bias_grad if not train else None, early_reduce)
        fused = self._token_block_enabled(pl) and hasattr(self, "tb_packed")
        if fused:
            # the dX chain of the transformer block, one workgroup per sample (csrc/token_block.hip): part 0 = final LN' .. dctx, the attention
            # backward, part 1 = dh.  The weight-gradient GEMMs read what the parts leave in HBM (df2, dg1 = df1, da1, dqkv, dr1) on the second stream.
            if "tb_part" not in b:
                b["tb_part"] = torch.empty(int(lib().eegclip_token_block_bwd_workspace_floats(B)), dtype=torch.float32, device=self.device)
            df2p, dg1p, da1p, dr1p, dqkvp = self._token_planes(b, B, "df2p", "dg1p", "da1p", "dr1p", "dqkvp")
            xp, hp, ctxp, n1p, g1p = self._token_planes(b, B, "xp", "hp", "ctxp", "n1p", "g1p")
            bd = _abi.TokenBlockBwdDesc(
                B=B, packed=_p(self.tb_packed), dn3=_p(b["dn3"]), n2=None, r2=_p(b["r2"]), r1=_p(b["r1"]), f1=_p(b["f1"]), mu1=_p(b["mu1"]),
                rs1=_p(b["rs1"]), mu2=_p(b["mu2"]), rs2=_p(b["rs2"]), mu3=_p(b["mu3"]), rs3=_p(b["rs3"]), ln1_g=_p(P[_LY + "norm1.weight"]),
                ln2_g=_p(P[_LY + "norm2.weight"]), ln2_b=_p(P[_LY + "norm2.bias"]), ln3_g=_p(P["encoder.encoder.norm.weight"]),
                dr1=_p(b["dr1"]), dctx=_p(b["dctx"]), partials=_p(b["tb_part"]), df2p=df2p, dg1p=dg1p, da1p=da1p, dr1p=dr1p, dqkvp=dqkvp,
                dln3_g=_p(G["encoder.encoder.norm.weight"]), dln3_b=_p(G["encoder.encoder.norm.bias"]), dln2_g=_p(G[_LY + "norm2.weight"]),
                dln2_b=_p(G[_LY + "norm2.bias"]), dln1_g=_p(G[_LY + "norm1.weight"]), dln1_b=_p(G[_LY + "norm1.bias"]),
                drop_p=pe_, seed=0, site_embed=SITE_EMBED, site_attn_out=SITE_ATTN_OUT, site_ffn_act=SITE_FFN_ACT, site_ffn_out=SITE_FFN_OUT)
            pl._keep.append(bd)
            if pe_ > 0.0:
                pl._seed_descs.append(bd)
            variant = 0                                                               # 512-thread workgroups (256-thread: 28.9 vs 25.0 us, csrc/wgrad_tok.hip)

            def wgrad_tok(tag, problems, side=True, slices=None, index=None):
                """the block's weight gradients from the token planes the fused kernels leave (csrc/wgrad_tok.hip): `problems` become ready together
                and run as ONE launch + one ordered slab reduction.  (name, dY planes, groups, X planes, M, N, heads_m, heads_n, bias, bias_mfma)"""
                arr = (_abi.WgradTokProblem * len(problems))()
                for i, (name, a, mg, x, M, N, hm, hn, bias, bm) in enumerate(problems):
                    arr[i] = _abi.WgradTokProblem(a=a, b=x, a_group_stride=B * 65536 if mg > 1 else 0, m_groups=mg, heads_m=hm, heads_n=hn, M=M, N=N,
                                                  out=_p(G[name]), ldo=N, bias_out=_p(G[bias]) if bias else None, bias_mfma=bm, sample_index=index)
                groups = sum(q[2] for q in problems)
                slices = slices or (wsk if wsk > 0 else int(lib().eegclip_wgrad_tok_slices(groups, B)))
                key = "wk:" + tag
                if key not in b:
                    b[key] = torch.empty(int(lib().eegclip_wgrad_tok_workspace_floats(arr, len(problems), B, slices)), dtype=torch.float32, device=self.device)
                pl._keep.append(arr)
                ops = (len(pl.ops), len(pl.ops) + 1)
                pl.call("eegclip_wgrad_tok", arr, len(problems), B, slices, _p(b[key]), variant, side=side)
                pl.call("eegclip_wgrad_tok_reduce", arr, len(problems), B, slices, _p(b[key]), side=side)
                return arr, ops

            # EEGCLIP_WGRAD_MERGE: 0 = three weight-gradient launches, each as soon as its dY planes exist (second stream);
            # 1 = the q | k | v gradient waits for the embedding's and shares its launch; 2 = all five in ONE launch after the last dY (joint-subject model: the
            # four shared ones in one launch, the per-subject value embeddings in theirs).  The fused backward
            # kernels own their CUs (160 KB of LDS, 2 waves per SIMD of 200+ VGPRs), so an "overlapping" weight-gradient launch in fact queues for CUs behind
            # them: fewer, fuller launches read the same 192 MB of planes with more workgroups in flight
            merge = int(os.environ.get("EEGCLIP_WGRAD_MERGE", "2"))
            late = []
            ffn_out = [
                (_LY + "conv2.weight", df2p, 1, g1p, D_MODEL, D_FF, 0, 0, _LY + "conv2.bias", 1),          # g1 has 256 real channels: no ones column
                (_LY + "conv1.weight", dg1p, 1, n1p, D_FF, D_MODEL, 0, 0, _LY + "conv1.bias", 0),
                (_LY + "attention.out_projection.weight", da1p, 1, ctxp, D_MODEL, HE, 0, 1, _LY + "attention.out_projection.bias", 0)]
            qkv_w = [(_LY + "attention.query_projection.weight", dqkvp, 3, hp, 3 * HE, D_MODEL, 1, 0, _LY + "attention.query_projection.bias", 0)]
            pl.call("eegclip_token_block_bwd", ctypes.byref(bd), 0)
            if defer_small:
                pl.deferred.append(lambda: pl.call("eegclip_token_block_bwd", ctypes.byref(bd), 2, side=True))
            else:
                pl.call("eegclip_token_block_bwd", ctypes.byref(bd), 2, side=ln_side)
            if merge >= 2:
                late += ffn_out
            else:
                wgrad_tok("ffn_out", ffn_out)
            pl.call(attn_bwd, _p(b["qkv"]), _p(b["dctx"]), dqkvp, 1, B, L_TOK, N_HEADS, D_HEAD, 3 * HE, 1.0 / math.sqrt(D_HEAD),
                    pe_, 0, SITE_ATTN, seed_at=11)
            if merge >= 1:
                late += qkv_w
            else:
                wgrad_tok("qkv", qkv_w)
            pl.call("eegclip_token_block_bwd", ctypes.byref(bd), 1)
        else:
            # final LN, LN2
            pl.call("eegclip_layernorm_bwd_full", _p(b["dn3"]), _p(b["n2"]), _p(P["encoder.encoder.norm.weight"]), _p(b["mu3"]), _p(b["rs3"]), _p(b["dn2"]),
                    _p(G["encoder.encoder.norm.weight"]), _p(G["encoder.encoder.norm.bias"]), R, D_MODEL, 0, None, 0.0, 0, 0, lnf_ws)
            # FFN: r2 = n1 + dropout(W2 dropout(gelu(W1 n1 + b1)) + b2).  LN2 backward emits dr2 (residual path) and df2 = dropout'(dr2);
            # bias gradients ride on the weight-gradient GEMMs; dropout' and gelu' of the hidden activation are the epilogue of the GEMM
            # that produces its gradient -- 5 elementwise / reduction passes over (B*64, 250..256) tensors gone
            pl.call("eegclip_layernorm_bwd_full", _p(b["dn2"]), _p(b["r2"]), _p(P[_LY + "norm2.weight"]), _p(b["mu2"]), _p(b["rs2"]), _p(b["dr2"]),
                    _p(G[_LY + "norm2.weight"]), _p(G[_LY + "norm2.bias"]), R, D_MODEL, 0, _p(b["df2"]), pe_, 0, SITE_FFN_OUT, lnf_ws, seed_at=13)
            wgrad(_LY + "conv2.weight", _p(b["df2"]), D_MODEL, _p(b["g1"]), D_FF, D_MODEL, D_FF, R, bias=_LY + "conv2.bias")
            pl.gemm(R, D_FF, D_MODEL, _p(b["df2"]), D(D_MODEL), D(1), _p(P[_LY + "conv2.weight"]), D(D_FF), D(1), _p(b["dg1"]), D(D_FF), D(1),
                    act=ACT_GELU_GRAD, R=_p(b["f1"]), Rm=D(D_FF), Rn=D(1), drop_p=pe_, drop_site=SITE_FFN_ACT)                  # dg1 := df1
            wgrad(_LY + "conv1.weight", _p(b["dg1"]), D_FF, _p(b["n1"]), D_MODEL, D_FF, D_MODEL, R, bias=_LY + "conv1.bias")
            pl.gemm(R, D_MODEL, D_FF, _p(b["dg1"]), D(D_FF), D(1), _p(P[_LY + "conv1.weight"]), D(D_MODEL), D(1), _p(b["dr2"]), D(D_MODEL), D(1),
                    accumulate=1)                                              # dr2 := dn1
            # attention block: r1 = h + dropout(Wo ctx + bo)
            pl.call("eegclip_layernorm_bwd_full", _p(b["dr2"]), _p(b["r1"]), _p(P[_LY + "norm1.weight"]), _p(b["mu1"]), _p(b["rs1"]), _p(b["dr1"]),
                    _p(G[_LY + "norm1.weight"]), _p(G[_LY + "norm1.bias"]), R, D_MODEL, 0, _p(b["da1"]), pe_, 0, SITE_ATTN_OUT, lnf_ws, seed_at=13)
            wgrad(_LY + "attention.out_projection.weight", _p(b["da1"]), D_MODEL, _p(b["ctx"]), HE, D_MODEL, HE, R,
                  bias=_LY + "attention.out_projection.bias")
            pl.gemm(R, HE, D_MODEL, _p(b["da1"]), D(D_MODEL), D(1), _p(P[_LY + "attention.out_projection.weight"]), D(HE), D(1), _p(b["dctx"]), D(HE), D(1))
            if attn_bwd == "eegclip_attention_bwd_x3":
                pl.call(attn_bwd, _p(b["qkv"]), _p(b["dctx"]), _p(b["dqkv"]), 0, B, L_TOK, N_HEADS, D_HEAD, 3 * HE, 1.0 / math.sqrt(D_HEAD),
                        pe_, 0, SITE_ATTN, seed_at=11)
            else:
                pl.call(attn_bwd, _p(b["qkv"]), _p(b["dctx"]), _p(b["dqkv"]), B, L_TOK, N_HEADS, D_HEAD, 3 * HE, 1.0 / math.sqrt(D_HEAD),
                        pe_, 0, SITE_ATTN, seed_at=10)
            wgrad(_LY + "attention.query_projection.weight", _p(b["dqkv"]), 3 * HE, _p(b["h"]), D_MODEL, 3 * HE, D_MODEL, R,
                  bias=_LY + "attention.query_projection.bias")                                    # q|k|v weights and biases are adjacent
            pl.gemm(R, D_MODEL, 3 * HE, _p(b["dqkv"]), D(3 * HE), D(1), _p(P[_LY + "attention.query_projection.weight"]), D(D_MODEL), D(1),
                    _p(b["dr1"]), D(D_MODEL), D(1), accumulate=2, drop_p=pe_, drop_site=SITE_EMBED)
            # ^ dr1 := dropout'(dh): the embedding dropout's backward (Embed.py:162) is the epilogue of the GEMM that completes dh -- accumulate FIRST
            #   (residual-path gradient already in dr1), then the mask of the (B,64,250) element index = m * 250 + n -- instead of a separate in-place
            #   pass over the 16 MB tensor (35 us in the step)
        # embedding: token row + value embedding
        tokg = G[_TOK_SHARED] if shared else G[_TOK_TABLE]
        # (drop_p = 0: the dropout' is already in dr1 -- this launch only sums the token rows' gradients, reads dr1 and can run beside the embedding's
        #  weight gradient)
        pl.call("eegclip_embed_finish_bwd", _p(b["dr1"]), _p(tokg), None if shared else _p(b["ids"]), B, L_TOK, D_MODEL, 0.0, 0, SITE_EMBED, side=True)
        for f in pl.deferred:             # (second-stream ops right behind one another share one fork: csrc/plan_exec.hip)
            f()
        hmap = D(D_MODEL, div=N_CH, so=L_TOK * D_MODEL)
        if want_dx:
            b["dx"] = torch.empty(B, N_CH, T_LEN, dtype=torch.float32, device=self.device)
        pl.x_gemm = None
        if fused:
            # value embedding: dY = the token-row gradients part 1 left as planes, X = the EEG sample planes of the forward (row 0 zero: the subject
            # token has no EEG row; ones column in rows 1..63: bias gradient = sum of the 63 channel rows of every sample)
            if not self.joint:
                wgrad_tok("embed" + ("+%d" % len(late) if late else ""), late + [(_E + "value_embedding.weight", dr1p, 1, xp, D_MODEL, T_LEN, 0, 0, _E + "value_embedding.bias", 0)], side=False)
            else:
                # joint-subject model: one problem per subject PRESENT in the batch, each contracting over that subject's samples only -- a range of
                # the subject-ordered sample list b["perm"] (the planes stay in batch order: the kernel looks the sample up per k-tile).  The member
                # problems, their count and sample ranges are patched per call (_joint_layout); absent subjects get no gradient (their .grad stays
                # None as in the reference, Embed.py:142-144).
                if late:
                    wgrad_tok("block", late, side=False)
                pl.j_wk_template = [(w, dr1p, 1, xp, D_MODEL, T_LEN, 0, 0, bk, 0) for w, bk in self.ve_keys]
                # K slices per subject so that (subjects present) x 4 tiles x slices fills the chip: 32 for one subject .. 6 for ten; the workspace takes
                # the largest product
                pl.j_wk_slices = [0] + [max(1, min(32, 2 * B // 4, (64 // n) // 8 * 8 if 64 // n >= 8 else 64 // n)) for n in range(1, self.n_subj + 1)]
                pl.j_wk_arr, pl.j_wk_ops = wgrad_tok("embed_joint", pl.j_wk_template, side=False, slices=8, index=_p(b["perm"]))
                need = max(int(lib().eegclip_wgrad_tok_workspace_floats(pl.j_wk_arr, n, B, pl.j_wk_slices[n])) for n in range(1, self.n_subj + 1))
                if b["wk:embed_joint"].numel() < need:
                    b["wk:embed_joint"] = torch.empty(need, dtype=torch.float32, device=self.device)
                    for op in pl.j_wk_ops:
                        pl.set_arg(op, 4, _p(b["wk:embed_joint"]))
                pl.j_wk_all = (_abi.WgradTokProblem * self.n_subj)(*pl.j_wk_arr)          # (the per-subject originals; j_wk_arr holds this batch's members)
                pl._keep.append(pl.j_wk_all)
        elif not self.joint:
            pl.x_gemm = pl.gemm(D_MODEL, T_LEN, B * N_CH, _p(b["dr1"]) + 4 * D_MODEL, D(1), hmap, 0, D(T_LEN), D(1),
                    _p(G[_E + "value_embedding.weight"]), D(T_LEN), D(1), accumulate=1, split_k=sk(B * N_CH),
                    rowsum_a=_p(G[_E + "value_embedding.bias"]))      # bias gradient = sum of the 63 channel rows of every sample
        if not self.joint:
            if want_dx:
                pl.gemm(B * N_CH, T_LEN, D_MODEL, _p(b["dr1"]) + 4 * D_MODEL, hmap, D(1),
                        _p(P[_E + "value_embedding.weight"]), D(T_LEN), D(1), _p(b["dx"]), D(T_LEN), D(1))
        elif fused and not want_dx:
            pass                                   # (the weight gradients above are all the joint-subject value embedding needs)
        else:
            # per-subject weight gradients over the subject-ordered batch (mirror of the forward: gather the token-row gradients into
            # subject order first when the batch is not; xs still holds the gathered EEG)
            XR = N_CH * T_LEN
            pl.j_gather = len(pl.ops)
            pl.call("eegclip_gather_rows", _p(b["hs"]) + 4 * D_MODEL, L_TOK * D_MODEL, _p(b["dr1"]) + 4 * D_MODEL, L_TOK * D_MODEL, _p(b["perm"]), B,
                    N_CH * D_MODEL, 0)
            if not fused:
                pl.j_gemm = [pl.desc(D_MODEL, T_LEN, N_CH, 0, D(1), hmap, 0, D(T_LEN), D(1), _p(G[w]), D(T_LEN), D(1), accumulate=1, split_k=1,
                                     rowsum_a=_p(G[bk])) for w, bk in self.ve_keys]
                pl.j_arr, pl.j_group = pl.gemm_grouped(self.n_subj)
            if want_dx:
                b["dxs"] = torch.empty(B, N_CH, T_LEN, dtype=torch.float32, device=self.device)
                pl.j_dx = [pl.desc(N_CH, T_LEN, D_MODEL, 0, hmap, D(1), _p(P[w]), D(T_LEN), D(1), 0, D(T_LEN), D(1)) for w, _ in self.ve_keys]
                pl.j_dx_arr, pl.j_dx_group = pl.gemm_grouped(self.n_subj)
                pl.j_scatter = len(pl.ops)
                pl.call("eegclip_gather_rows", _p(b["dx"]), XR, _p(b["dxs"]), XR, _p(b["perm"]), B, XR, 1)
        return pl

    def _build_bwd_head_f32(self, pl, b, B, pp_, ln_side, wgrad):
        """the projection head's backward on fp32-operand GEMMs (exact-fp32 plans, EEGCLIP_HEAD_GEMM=0): split-K with atomics into the cleared arena"""
        P, G = self.P, self.G
        # s = u + dropout(W4 gelu(u) + b4): the LayerNorm backward writes ds and dv = ds * mask / (1 - p) in one pass
        pl.call("eegclip_layernorm_bwd", 0, _p(b["s"]), _p(P["proj_eeg.2.weight"]), _p(b["mu4"]), _p(b["rs4"]), _p(b["ds"]),
                None, None, B, P_DIM, 0, _p(b["dv"]), pp_, 0, SITE_PROJ, seed_at=13)
        # (the gamma / beta gradients of every LayerNorm are a second, independent kernel: it runs on the side stream, off the dX chain)
        pl.dout_par_op = len(pl.ops)
        pl.call("eegclip_layernorm_bwd", 0, _p(b["s"]), None, _p(b["mu4"]), _p(b["rs4"]), None,
                _p(G["proj_eeg.2.weight"]), _p(G["proj_eeg.2.bias"]), B, P_DIM, 0, None, 0.0, 0, 0, side=ln_side)
        wgrad("proj_eeg.1.fn.1.weight", _p(b["dv"]), P_DIM, _p(b["gu"]), P_DIM, P_DIM, P_DIM, B, bias="proj_eeg.1.fn.1.bias")
        skh = _head_split(B)
        pl.gemm(B, P_DIM, P_DIM, _p(b["dv"]), D(P_DIM), D(1), _p(P["proj_eeg.1.fn.1.weight"]), D(P_DIM), D(1), _p(b["dgu"]), D(P_DIM), D(1),
                accumulate=int(skh > 1), split_k=skh)
        pl.call("eegclip_gelu_bwd", _p(b["dgu"]), _p(b["u"]), _p(b["ds"]), B * P_DIM, 1, 0.0, 0, 0)          # ds := du
        wgrad("proj_eeg.0.weight", _p(b["ds"]), P_DIM, _p(b["feat"]), F_TS, P_DIM, F_TS, B, bias="proj_eeg.0.bias")
        pl.gemm(B, F_TS, P_DIM, _p(b["ds"]), D(P_DIM), D(1), _p(P["proj_eeg.0.weight"]), D(F_TS), D(1), _p(b["dfeat"]), D(F_TS), D(1),
                accumulate=int(skh > 1), split_k=skh)

    def _build_bwd_cstack(self, pl, b, B, train, W, zsum, conv_bias_grad, early_reduce, defer_small=False):
        """spatial conv + BN1 + ELU + temporal conv backward recomputed from the token rows (csrc/cstack_bwd.hip, round 5): y1 / z1 / dz1 / dy1 never
        exist in HBM.  dWs on the second stream; BatchNorm1-backward sums as one partial row per sample, summed in a fixed order by the apply pass."""
        P, G, sums, bn = self.P, self.G, b["sums"], b["bn"]
        dev = self.device
        if "csw_ws" not in b:
            b["csw_ws"] = torch.empty(int(lib().eegclip_cstack_bwd_w2_workspace_floats(B, N_CH)), dtype=torch.float32, device=dev)
            b["csb_ws"] = torch.empty(int(lib().eegclip_cstack_bwd_workspace_floats(B)), dtype=torch.float32, device=dev)
            b["cs_rows3"] = torch.empty(B, 2 * C_TS, dtype=torch.float64, device=dev)
        xs = (_p(b["n3"]), L_TOK * D_MODEL, D_MODEL)
        bnp = (_p(bn[0]), _p(bn[1]), _p(P[_TS + "2.weight"]), _p(P[_TS + "2.bias"]))
        pl.call("eegclip_cstack_bwd_w2", *xs, _p(P[_TS + "0.weight"]), _p(P[_TS + "0.bias"]), *bnp, _p(b["dy2"]), _p(G[_TS + "4.weight"]), _p(b["csw_ws"]),
                B, N_CH, side=os.environ.get("EEGCLIP_W2_SIDE", "1") != "0")
        rows3 = b["cs_rows3"]
        count1 = float(W * B * N_CH * W_TS)
        common = dict(B=B, H=N_CH, x=xs[0], xs_b=xs[1], xs_h=xs[2], w25=_p(P[_TS + "0.weight"]), bias1=_p(P[_TS + "0.bias"]), mean1=bnp[0], rstd1=bnp[1],
                      gamma1=bnp[2], beta1=bnp[3], packed_t=_p(self.cs_packed_t), dy2=_p(b["dy2"]), rows_out=_p(rows3), count=count1,
                      dgamma=_p(G[_TS + "2.weight"]), dbeta=_p(G[_TS + "2.bias"]), dx=_p(b["dn3"]), dw_partials=_p(b["csb_ws"]), dw25=_p(G[_TS + "0.weight"]))
        pl.call_desc("eegclip_cstack_bwd_stats", _abi.CstackBwdDesc(stat=None, nstat=0, stat_local=None, nstat_local=0, **common))
        stat, nstat, local, nlocal = _p(rows3), B, None, 0
        if W > 1 or not train:
            pl.call("eegclip_bn_finalize_rows", _p(rows3), B, count1, EPS, 0.1, C_TS, None, None, None, None, None, _p(sums[3]))
            local, nlocal = _p(rows3), B
            if W > 1:
                pl.callback(lambda: self._allreduce(sums[3]), "allreduce_bn1_bwd")
            stat, nstat = (_p(sums[3]), 1) if train else (_p(zsum), 1)
            if not train:
                pl.callback(conv_bias_grad(_TS + "0.bias", _TS + "2.weight", bn[1], sums[3]), "conv1_bias_grad_eval")
        # (the tap gradient leaves the apply pass as one partial row per sample; their sum is read by the optimizer only: second stream)
        pl.call_desc("eegclip_cstack_bwd_apply", _abi.CstackBwdDesc(stat=stat, nstat=nstat, stat_local=local, nstat_local=nlocal, **dict(common, dw25=None)))
        pl.early_cut = len(pl.ops)          # every gradient of the early bucket (loss scale, conv stack, head) has been ISSUED once the ops before this index
        #                                     and the taps reduction below have: the step plan starts their optimizer update here (step_plan.py)

        def taps():
            pl.taps_op = len(pl.ops)
            pl.call("eegclip_cstack_bwd_taps_reduce", _p(b["csb_ws"]), B, _p(G[_TS + "0.weight"]), side=True)
        if early_reduce or not defer_small:
            taps()
        else:
            pl.deferred.append(taps)          # with the other small reductions, behind ONE fork at the end of the backward (see _build_bwd)
        if early_reduce:
            # every gradient of the conv stack and the head is final here: start their all-reduce now (asynchronously, ordered behind both
            # streams); dist.average_flat_grads() waits for it after the backward and reduces the rest
            pl.callback(self._start_early_reduce, "allreduce_early_bucket", side=True)

    def _build_bwd_conv_y1(self, pl, b, B, train, W, zsum, conv_bias_grad, early_reduce):
        """spatial conv + BN1 + ELU + temporal conv backward around y1 / dy1 in HBM (rounds 1-4 kernels; exact-fp32 plans only)"""
        P, G, sums, bn = self.P, self.G, b["sums"], b["bn"]
        # spatial conv + BN1 + ELU backward, fused around y1 (csrc/sconv.hip): dWs from re-evaluated z1; dz1 = Ws^T dy2 recomputed on the
        # matrix cores in both BatchNorm-backward passes instead of being written and re-read.
        # (d(conv bias) in front of a train-mode BatchNorm is identically zero: tsconv.0.bias / tsconv.4.bias keep the cleared zero.)
        if "scw_ws" not in b:
            b["scw_ws"] = torch.empty(int(lib().eegclip_sconv_bwd_w_workspace_floats(B, N_CH)), dtype=torch.float32, device=self.device)
        bnp = (_p(bn[0]), _p(bn[1]), _p(P[_TS + "2.weight"]), _p(P[_TS + "2.bias"]))
        wt = (None, None)                       # exact fp32 products (these plans only exist under EEGCLIP_GEMM_PRECISION=f32 since round 6)
        if "scx_ws" not in b:
            b["scx_ws"] = torch.empty(int(lib().eegclip_sconv_bwd_x_stats_workspace_floats(B)) // 2, dtype=torch.float64, device=self.device)
        # (folding the two into one pass over y1 -- round 3's eegclip_sconv_bwd_w_stats -- saved 15 us of kernel time but moved the weight gradient from
        #  the second stream onto the dX chain: step 1.100-1.108 vs 1.093-1.096 ms; removed in round 4)
        pl.call("eegclip_sconv_bwd_w", _p(b["y1"]), *bnp, _p(b["dy2"]), _p(G[_TS + "4.weight"]), _p(b["scw_ws"]), B, N_CH, pl.precision & 0xff, side=True)
        pl.call("eegclip_sconv_bwd_x_stats", _p(b["dy2"]), _p(P[_TS + "4.weight"]), *wt, _p(b["y1"]), *bnp, _p(sums[3]), _p(b["scx_ws"]), B, N_CH)
        local1 = None
        if W > 1:
            local1 = torch.zeros_like(sums[3])
            pl._keep.append(local1)

            def exchange1():
                local1.copy_(sums[3])
                self._allreduce(sums[3])
            pl.callback(exchange1, "allreduce_bn1_bwd")
        if not train:
            pl.callback(conv_bias_grad(_TS + "0.bias", _TS + "2.weight", bn[1], sums[3]), "conv1_bias_grad_eval")
        count1 = float(W * B * N_CH * W_TS)
        sums1 = (_p(sums[3]) if train else _p(zsum), (_p(local1) if local1 is not None else None) if train else _p(sums[3]))
        if "dy1" not in b:
            b["dy1"] = torch.empty(B, C_TS, N_CH, W_TS, dtype=torch.float32, device=self.device)
        pl.call("eegclip_sconv_bwd_x_apply", _p(b["dy2"]), _p(P[_TS + "4.weight"]), *wt, _p(b["y1"]), *bnp, *sums1, count1, _p(b["dy1"]),
                _p(G[_TS + "2.weight"]), _p(G[_TS + "2.bias"]), B, N_CH)
        if "tsw_ws" not in b:
            b["tsw_ws"] = torch.empty(int(lib().eegclip_tsconv_bwd_w_workspace_floats(B, N_CH)), dtype=torch.float32, device=self.device)
        pl.call("eegclip_tsconv_bwd_w", _p(b["n3"]), L_TOK * D_MODEL, D_MODEL, _p(b["dy1"]), _p(G[_TS + "0.weight"]), _p(b["tsw_ws"]), B, N_CH, T_LEN,
                C_TS, side=True)
        if early_reduce:
            # every gradient of the conv stack and the head is final here: start their all-reduce now (asynchronously, ordered behind both
            # streams); dist.average_flat_grads() waits for it after the backward and reduces the rest
            pl.callback(self._start_early_reduce, "allreduce_early_bucket", side=True)
        pl.call("eegclip_tsconv_bwd_x", _p(b["dy1"]), _p(P[_TS + "0.weight"]), _p(b["dn3"]), L_TOK * D_MODEL, D_MODEL, B, N_CH, T_LEN, C_TS)

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size() if (self.model.sync_batchnorm and dist.is_available() and dist.is_initialized()) else 1

    @staticmethod
    def _allreduce(t):
        import torch.distributed as dist
        dist.all_reduce(t)

    def _start_early_reduce(self):
        import torch.distributed as dist
        a0, a1 = self.early_bucket
        self.early_work = dist.all_reduce(self.gflat[a0:a1], op=dist.ReduceOp.SUM, async_op=True)

    # ---- execution -----------------------------------------------------------------------------------------------
    def _joint_layout(self, pl, b, B, host_ids, x_ptr, backward):
        """Point the per-subject GEMMs of a joint-model plan at this batch: subject blocks of the subject-ordered batch (the batch itself if its
        ids are already non-decreasing, else the gathered copy xs / hs)."""
        fused = getattr(pl, "tb_desc", None) is not None or hasattr(pl, "j_wk_arr") or getattr(pl, "j_fused", False)
        if not backward:
            a = np.asarray(host_ids, dtype=np.int64)
            in_order = bool((a[1:] >= a[:-1]).all())
            if fused:
                # the fused block takes each sample's value embedding from its subject id; the backward's per-subject weight gradients walk the
                # subject-ordered sample list (identity when the batch is already ordered).  Both go up in ONE asynchronous copy from a pinned
                # staging buffer (a ring: the copy of step i may still be in flight when step i + 1 fills the next slot) -- a blocking copy_ from
                # pageable memory would make the host wait for everything queued on the stream, every step
                perm = np.argsort(a, kind="stable")
                on_gpu = b["jmeta"].is_cuda
                if "jmeta_ring" not in b:                   # (not setdefault: its default is evaluated -- eight pinned allocations -- on every call)
                    b["jmeta_ring"] = [[torch.empty(2 * B, dtype=torch.int32).pin_memory() if on_gpu else torch.empty(2 * B, dtype=torch.int32), None]
                                       for _ in range(8)]
                ring = b["jmeta_ring"]
                slot = ring[b.get("jmeta_i", 0) % len(ring)]
                b["jmeta_i"] = b.get("jmeta_i", 0) + 1
                if slot[1] is not None:
                    slot[1].synchronize()                   # (the copy that last read this slot, 8 steps ago: long complete unless the host runs far ahead)
                stage = slot[0].numpy()
                stage[:B], stage[B:] = a, perm
                b["jmeta"].copy_(slot[0], non_blocking=True)
                if on_gpu:
                    slot[1] = slot[1] or torch.cuda.Event()
                    slot[1].record(current_stream())
                a = a[perm]
            elif not in_order:
                perm = np.argsort(a, kind="stable")
                b["perm"].copy_(torch.from_numpy(perm.astype(np.int32)))
                a = a[perm]
            # (subject s occupies positions [starts[s], starts[s] + counts[s]) of the subject-ordered batch)
            counts = np.bincount(a, minlength=max(self.n_subj, 1))
            starts = np.cumsum(counts) - counts
            b["segs"] = [(s_, st, c) for s_, (st, c) in enumerate(zip(starts.tolist(), counts.tolist())) if c]
            b["in_order"] = in_order
        segs, in_order = b["segs"], b["in_order"]
        XR, HR = 4 * N_CH * T_LEN, 4 * L_TOK * D_MODEL
        if fused and not backward:
            pl.tb_desc.x = x_ptr
            return
        skip = set()
        has_dx = hasattr(pl, "j_dx")
        has_gemm = hasattr(pl, "j_gemm")
        if hasattr(pl, "j_wk_arr"):
            for i, (s_, st, n) in enumerate(segs):
                q = pl.j_wk_all[s_]
                q.sample0, q.samples = st, n
                pl.j_wk_arr[i] = q                           # struct copy: the members of this batch, in subject order
            for op in pl.j_wk_ops:
                pl.set_arg(op, 1, len(segs))
                pl.set_arg(op, 3, pl.j_wk_slices[len(segs)])
            if not has_dx:
                return
        if in_order:
            skip.add(pl.j_gather)
            if not backward or has_dx:
                skip.add(pl.j_scatter)
        if not backward:
            pl.set_arg(pl.j_gather, 2, x_ptr)
            xb, hb = (x_ptr, _p(b["h"])) if in_order else (_p(b["xs"]), _p(b["hs"]))
            for s, st, n in segs:
                d = pl.j_gemm[s]
                d.M, d.A, d.C = n * N_CH, xb + st * XR, hb + st * HR + 4 * D_MODEL
        else:
            xb, gb = (x_ptr, _p(b["dr1"])) if in_order else (_p(b["xs"]), _p(b["hs"]))
            for s, st, n in segs:
                if has_gemm:
                    d = pl.j_gemm[s]
                    d.K, d.A, d.B = n * N_CH, gb + st * HR + 4 * D_MODEL, xb + st * XR
                    d.split_k = max(1, min(16, n * N_CH // 256))
                if has_dx:
                    d = pl.j_dx[s]
                    d.M, d.A, d.C = n * N_CH, gb + st * HR + 4 * D_MODEL, (_p(b["dx"]) if in_order else _p(b["dxs"])) + st * XR
        if has_gemm:
            for i, (s, _, _) in enumerate(segs):
                pl.j_arr[i] = pl.j_gemm[s]                   # struct copy: the members of this batch, in subject order
            pl.set_arg(pl.j_group, 1, len(segs))
        if has_dx:
            for i, (s, _, _) in enumerate(segs):
                pl.j_dx_arr[i] = pl.j_dx[s]
            pl.set_arg(pl.j_dx_group, 1, len(segs))
        pl.skip = frozenset(skip)

    def forward(self, x, ids, shared, train, host_ids=None):
        B = x.shape[0]
        probs = self.model.drop_probs(train)
        if B not in self.bufs:
            self.bufs[B] = self._alloc(B)
        key = (B, train, shared, probs, self._world())
        pk = ("f",) + key
        if pk not in self.plans:
            self.plans[pk] = self._build_fwd(B, train, shared, probs)
        pl = self.plans[pk]
        b = self.bufs[B]
        if not shared:
            uid = getattr(ids, "_eegclip_uniform_id", None)
            if uid is None or b.get("ids_uniform") != uid:            # (a single-subject loop sends the same ids every step: copy them once)
                b["ids"].copy_(ids)
                b["ids_uniform"] = uid
        if self.joint:
            self._joint_layout(pl, b, B, host_ids, x.data_ptr(), False)
        elif pl.tb_desc is not None:
            pl.tb_desc.x = x.data_ptr()
        else:
            pl.x_gemm.A = x.data_ptr()          # the only per-call pointer: the EEG batch itself
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if train and max(probs) > 0 else 0
        b["seed"] = seed
        out = torch.empty(B, P_DIM, dtype=torch.float32, device=self.device)
        pl.set_arg(pl.out_op, 8, out.data_ptr())
        pl.run(raw_stream(), seed)
        if getattr(pl, "clears_zb", False):
            b["zb_clean"] = True
        self.last_key = key
        self.version[B] = self.version.get(B, 0) + 1
        return out

    def attach_grads(self, shared, subjects=()):
        """Make p.grad views of the flat gradient buffer for every parameter that receives a gradient; zero the
        buffer if the optimizer cleared the grads (zero_grad(set_to_none=True) is torch's default)."""
        lk = (shared, tuple(sorted(subjects)) if self.joint else (), _dp_world() > 1 if self.joint else False)
        cached = self._live_cache.get(lk)
        if cached is None:
            live = self.live_base + [_TOK_SHARED if shared else _TOK_TABLE]
            if self.joint:
                everyone = lk[2]               # ranks must step the same parameters
                live = live + [k for s in (range(self.n_subj) if everyone else sorted(subjects)) for k in self.ve_keys[s]]
            if len(self._live_cache) > 256:
                self._live_cache.clear()
            # (parameter, its gradient view) pairs, once per live set: this runs on every backward
            cached = self._live_cache[lk] = (tuple(live), [(self.params[k], self.G[k]) for k in live])
        live, pairs = cached
        # one pass: mine (already the flat-buffer view) | None | foreign (written by somebody else before this backward ran)
        n_mine, foreign, rest = 0, [], []
        for p, g in pairs:
            pg = p.grad
            if pg is g or (pg is not None and pg.data_ptr() == g.data_ptr()):
                n_mine += 1
            else:
                rest.append((p, g))
                if pg is not None:
                    foreign.append((g, pg))
        if not rest:
            self._clear_for = None
            return False                                   # accumulating onto existing gradients
        if n_mine == 0:
            if self._clear_for != live:                    # (an optimizer step that cleared exactly these gradients behind its reads: nothing to do)
                self.gflat.zero_()                         # the common case after optimizer.zero_grad(): one memset
            elif _Engine.check_cleared and bool(self.gflat.any()):
                # debug (tests set _Engine.check_cleared; a host sync): skipping the clear is only sound if every accumulating backward kernel wrote
                # exclusively inside the views the optimizer cleared and nothing touched the buffer since
                raise EegclipError("flat gradient buffer is not clear although the optimizer reported clearing it")
        else:
            for p, g in rest:
                g.zero_()
        self._clear_for = None
        for g, pg in foreign:                              # e.g. logit_scale.grad written by the loss before this backward ran
            g.copy_(pg)
        for p, g in rest:
            p.grad = g
        self._attached = live
        return True

    def grads_cleared(self, grad_ptrs):
        """an optimizer step (optim.AdamW.step(zero_grad=True)) has zeroed the gradients at these addresses behind its reads: if they are exactly
        the views attached by the last backward, the next attach_grads() need not clear the flat buffer again"""
        att = getattr(self, "_attached", None)
        if att:
            mine = self._attached_ptrs.get(att)
            if mine is None:
                if len(self._attached_ptrs) > 256:
                    self._attached_ptrs.clear()
                mine = self._attached_ptrs[att] = frozenset(self.G[k].data_ptr() for k in att)
            if mine <= (grad_ptrs if isinstance(grad_ptrs, (set, frozenset)) else set(grad_ptrs)):
                self._clear_for = att

    def backward(self, key, x, dout, want_dx):
        B, train, shared, probs, W = key
        b = self.bufs[B]
        if "ds" not in b:
            self._alloc_bwd(B, b)
        early = bool(getattr(self.model, "overlap_grad_allreduce", False)) and _dp_world() > 1
        pk = ("b", B, train, shared, probs, want_dx, W, early)
        if pk not in self.plans:
            self.plans[pk] = self._build_bwd(B, shared, probs, want_dx, early, train)
        pl = self.plans[pk]
        self.attach_grads(shared, {s for s, _, _ in b["segs"]} if self.joint else ())
        pl.set_arg(pl.dout_op, 0, dout.data_ptr())
        pl.set_arg(pl.dout_par_op, 0, dout.data_ptr())
        pl._keep_x = (x, dout)
        if self.joint:
            self._joint_layout(pl, b, B, None, x.data_ptr(), True)
        elif pl.x_gemm is not None:
            pl.x_gemm.B = x.data_ptr()          # the value-embedding weight-gradient GEMM reads the EEG batch
        if not getattr(pl, "lean", False):
            if not b["zb_clean"]:               # an eval-mode forward, or a second backward through one forward: clear the arena here
                b["zb"].zero_()
            b["zb_clean"] = False
        pl.run(raw_stream(), b.get("seed", 0))
        return b["dx"].clone() if want_dx else None
