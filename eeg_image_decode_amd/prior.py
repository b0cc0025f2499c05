"""Diffusion prior (EEG embedding -> CLIP image embedding) on MI355X with the reference's surface
(Generation/diffusion_prior.py): DiffusionPriorUNet, EmbeddingDataset, Pipe.train / Pipe.generate, plus the DDPM
scheduler pieces the reference takes from diffusers==0.30.0 (restated: diffusers is not a dependency here).

* DiffusionPriorUNet keeps the reference's state_dict keys (input_layer.0/1, encode_time_embedding.i.linear_1/2, ...,
  output_layer); parameters / gradients are views of flat buffers; forward and backward are replayed launch plans.  Training at batches that
  are multiples of 64 (round 4): every Linear is a plane GEMM (csrc/gemm_planes.hip: operands and results as bf16 hi | lo planes, split-bf16
  products), weight gradients through eegclip_wgrad_planes, fused stage tails (csrc/prior.hip); other shapes / exact-fp32 arithmetic: the
  general GEMM (SiLU / residual / accumulate epilogues) + LayerNorm->SiLU->dropout kernels.
* Pipe.train reproduces the reference step order: 10 % whole-batch condition drop, epsilon-MSE, clip_grad_norm_(1.0),
  LR scheduler stepped BEFORE the optimizer, Adam.  Loss is accumulated on the device (one host sync per epoch); the
  gradient-norm clip factor stays on the device (no sync per step).
* Pipe.generate is batched (the reference is batch-1 only because of `t.long().item()`): each of the N rows is an
  independent chain; with a seeded CPU generator it reproduces the reference's noise stream exactly.
"""
import math
import os

import torch
import torch.nn as nn
from torch.utils.data import Dataset

from . import _abi
from ._lib import EegclipError, check, lib, raw_stream, require_cuda
from .plan import Plan, default_gemm_precision

D = _abi.dim
ACT_SILU = _abi.ACT_SILU


def _p(t):
    return t.data_ptr()


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise EegclipError(f"{type(self).__name__} holds parameters only; call DiffusionPriorUNet.forward (HIP kernels)")


class Timesteps(_Holder):
    def __init__(self, num_channels, flip_sin_to_cos=True, downscale_freq_shift=0):
        super().__init__()
        if not flip_sin_to_cos or downscale_freq_shift != 0:
            raise EegclipError("only Timesteps(dim, True, 0) (the reference's configuration) is implemented")
        self.num_channels = num_channels


class TimestepEmbedding(_Holder):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class DiffusionPriorUNet(nn.Module):
    def __init__(self, embed_dim=1024, cond_dim=42, hidden_dim=[1024, 512, 256, 128, 64], time_embed_dim=512, act_fn=nn.SiLU, dropout=0.0):
        super().__init__()
        if act_fn is not nn.SiLU:
            raise EegclipError("the HIP stage kernel fuses LayerNorm->SiLU->dropout; act_fn must be nn.SiLU (the reference default)")
        if max(hidden_dim) > 1024 or embed_dim > 1024:
            raise EegclipError("LayerNorm kernel handles rows of <= 1024 features")
        self.embed_dim, self.cond_dim, self.hidden_dim = embed_dim, cond_dim, list(hidden_dim)
        self.time_embed_dim = time_embed_dim
        self.time_proj = Timesteps(time_embed_dim, True, 0)
        h = self.hidden_dim
        self.input_layer = nn.Sequential(nn.Linear(embed_dim, h[0]), nn.LayerNorm(h[0]), act_fn())
        self.num_layers = len(h)
        n = self.num_layers
        self.encode_time_embedding = nn.ModuleList([TimestepEmbedding(time_embed_dim, h[i]) for i in range(n - 1)])
        self.encode_cond_embedding = nn.ModuleList([nn.Linear(cond_dim, h[i]) for i in range(n - 1)])
        self.encode_layers = nn.ModuleList([nn.Sequential(nn.Linear(h[i], h[i + 1]), nn.LayerNorm(h[i + 1]), act_fn(), nn.Dropout(dropout)) for i in range(n - 1)])
        self.decode_time_embedding = nn.ModuleList([TimestepEmbedding(time_embed_dim, h[i]) for i in range(n - 1, 0, -1)])
        self.decode_cond_embedding = nn.ModuleList([nn.Linear(cond_dim, h[i]) for i in range(n - 1, 0, -1)])
        self.decode_layers = nn.ModuleList([nn.Sequential(nn.Linear(h[i], h[i - 1]), nn.LayerNorm(h[i - 1]), act_fn(), nn.Dropout(dropout)) for i in range(n - 1, 0, -1)])
        self.output_layer = nn.Linear(h[0], embed_dim)
        self._eng = None

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self._eng = None
        return r

    def _engine(self):
        if self._eng is None or self._eng.stale():
            require_cuda(self.output_layer.weight.data, "DiffusionPriorUNet parameters (call .cuda() first)")
            self._eng = _PriorEngine(self)
        return self._eng

    def flat_parameters(self):
        e = self._engine()
        return e.flat, e.gflat

    def drop_p(self):
        ps = {float(l[3].p) for l in list(self.encode_layers) + list(self.decode_layers)}
        if len(ps) != 1:
            raise EegclipError("all stage dropouts must share one p")
        return ps.pop() if self.training else 0.0

    def forward(self, x, t, c=None):
        """x (N,embed) f32, t (N,) int or float timesteps, c (N,cond) f32 or None -> predicted noise (N,embed)."""
        eng = self._engine()
        require_cuda(x, "x")
        x = x.float().contiguous()
        t = t.to(device=x.device, dtype=torch.float32).contiguous()
        if c is not None:
            c = c.to(x.device).float().contiguous()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _PriorFn.apply(x, eng.anchor, self, t, c)
        return eng.forward(x, t, c, self.drop_p()).clone()


class _PriorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, model, t, c):
        eng = model._engine()
        out = eng.forward(x, t, c, model.drop_p())
        ctx.eng, ctx.key, ctx.version = eng, eng.last_key, eng.version[eng.last_key]
        ctx.saved = (x, t, c)
        return out.clone()

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        if eng.version.get(ctx.key) != ctx.version:
            raise EegclipError("prior activations were overwritten by a later forward at the same batch size")
        x, t, c = ctx.saved
        eng.backward(ctx.key, x, c, dout.contiguous())
        return None, None, None, None, None


class _PriorEngine:
    def __init__(self, model):
        self.model = model
        params = dict(model.named_parameters())
        dev = model.output_layer.weight.device
        self.device = dev
        h = model.hidden_dim
        n = model.num_layers
        # stage table: (prefix_time, prefix_cond, prefix_layer, h_in, h_out, skip_from / skip_to bookkeeping)
        self.stages = []
        for i in range(n - 1):
            self.stages.append(dict(t=f"encode_time_embedding.{i}.", c=f"encode_cond_embedding.{i}.", l=f"encode_layers.{i}.", hin=h[i], hout=h[i + 1], dec=None))
        for j, i in enumerate(range(n - 1, 0, -1)):
            self.stages.append(dict(t=f"decode_time_embedding.{j}.", c=f"decode_cond_embedding.{j}.", l=f"decode_layers.{j}.", hin=h[i], hout=h[i - 1], dec=j))
        # The first Linear of every stage's time embedding reads the same input (the sinusoid of t), and so does every stage's condition
        # Linear (c): stored back to back in stage order, each group is ONE weight matrix (sum of stage widths x 512 / x cond_dim) and runs
        # as one GEMM in training -- 2 launches instead of 16 forward, and one weight-gradient GEMM instead of 8 for the time group.
        self.wide = sum(st["hin"] for st in self.stages)
        self.col0 = [sum(st["hin"] for st in self.stages[:i]) for i in range(len(self.stages))]
        groups = [[st["t"] + "linear_1.weight" for st in self.stages], [st["t"] + "linear_1.bias" for st in self.stages],
                  [st["c"] + "weight" for st in self.stages], [st["c"] + "bias" for st in self.stages]]
        grouped = [k for g in groups for k in g]
        if any(params[k].numel() % 4 for k in grouped):
            raise EegclipError("hidden_dim entries must be multiples of 4 (grouped parameter blocks are stored without padding)")
        order = grouped + [k for k in params if k not in set(grouped)]
        offs, off = {}, 0
        for k in order:
            offs[k] = off
            off += (params[k].numel() + 3) // 4 * 4
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)
        self.P, self.G, self.params = {}, {}, params
        for k, p in params.items():
            v = self.flat[offs[k]:offs[k] + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            self.P[k] = v
            self.G[k] = self.gflat[offs[k]:offs[k] + p.numel()].view(p.shape)
        self._first = params[order[0]]
        self.cond_keys = [k for k in params if "cond_embedding" in k]
        self.bufs, self.plans, self.version = {}, {}, {}
        self.last_key = None
        lib()

    def stale(self):
        return self._first.data_ptr() != self.flat.data_ptr()

    def _alloc(self, N):
        dev, m = self.device, self.model
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        h0 = m.hidden_dim[0]
        b = dict(temb=f(N, m.time_embed_dim), linI=f(N, h0), lnI=f(N, h0), muI=f(N), rsI=f(N), actI=f(N, h0), out=f(N, m.embed_dim),
                 tt=f(N), dout=f(N, m.embed_dim), dactI=f(N, h0), dlnI=f(N, h0), dlinI=f(N, h0))
        # stage s owns columns col0[s] .. col0[s] + h_in of these (N, wide) buffers (row stride = wide): time-embedding hidden activations,
        # the stage inputs x + t_emb + c_emb, and the gradient of the time-embedding hidden layer
        W = self.wide
        b.update(T1pre=f(N, W), T1act=f(N, W), XIN=f(N, W), DT1=f(N, W))
        for s, st in enumerate(self.stages):
            hi, ho = st["hin"], st["hout"]
            b.update({f"lin{s}": f(N, ho), f"ln{s}": f(N, ho), f"mu{s}": f(N),
                      f"rs{s}": f(N), f"act{s}": f(N, ho), f"dact{s}": f(N, ho), f"dln{s}": f(N, ho), f"dlin{s}": f(N, ho), f"dxin{s}": f(N, hi)})
        return b

    def _build_fwd(self, N, cond, p, cond_rows=None):
        P, b, m = self.P, self.bufs[N], self.model
        pl = Plan(f"prior_fwd[N={N}]")
        E, Td, Cd, h0 = m.embed_dim, m.time_embed_dim, m.cond_dim, m.hidden_dim[0]
        pl.call("eegclip_timestep_embedding", _p(b["tt"]), N, Td, _p(b["temb"]))
        pl.x_gemm = pl.gemm(N, h0, E, 0, D(E), D(1), _p(P["input_layer.0.weight"]), D(1), D(E), _p(b["linI"]), D(h0), D(1), bias_n=_p(P["input_layer.0.bias"]))
        pl.call("eegclip_layernorm_silu_fwd", _p(b["linI"]), _p(P["input_layer.1.weight"]), _p(P["input_layer.1.bias"]), _p(b["lnI"]), _p(b["actI"]),
                _p(b["muI"]), _p(b["rsI"]), N, h0, 1e-5, 0.0, 0, 0)
        pl.c_gemms = []
        cur = "actI"
        n_enc = m.num_layers - 1
        skips = []
        W, s0 = self.wide, self.stages[0]
        # all eight time-embedding hidden layers: SiLU(t_emb W1^T + b1) -> T1act (pre-activation kept for the backward)
        pl.gemm(N, W, Td, _p(b["temb"]), D(Td), D(1), _p(P[s0["t"] + "linear_1.weight"]), D(1), D(Td), _p(b["T1act"]), D(W), D(1),
                Cpre=_p(b["T1pre"]), bias_n=_p(P[s0["t"] + "linear_1.bias"]), act=ACT_SILU)
        if cond:
            # all eight condition embeddings c Wc^T + bc -> XIN (the stage inputs accumulate onto them)
            rows = N if cond_rows is None else cond_rows
            if rows < N:
                pl.memset(b["XIN"])
            pl.c_gemms.append(pl.gemm(rows, W, Cd, 0, D(Cd), D(1), _p(P[s0["c"] + "weight"]), D(1), D(Cd), _p(b["XIN"]), D(W), D(1),
                                      bias_n=_p(P[s0["c"] + "bias"])))
        for s, st in enumerate(self.stages):
            hi, ho = st["hin"], st["hout"]
            if st["dec"] is None:
                skips.append(cur)
            xin = _p(b["XIN"]) + 4 * self.col0[s]
            pl.gemm(N, hi, hi, _p(b["T1act"]) + 4 * self.col0[s], D(W), D(1), _p(P[st["t"] + "linear_2.weight"]), D(1), D(hi), xin, D(W), D(1),
                    bias_n=_p(P[st["t"] + "linear_2.bias"]), R=_p(b[cur]), Rm=D(hi), Rn=D(1), accumulate=1 if cond else 0)
            pl.gemm(N, ho, hi, xin, D(W), D(1), _p(P[st["l"] + "0.weight"]), D(1), D(hi), _p(b[f"lin{s}"]), D(ho), D(1),
                    bias_n=_p(P[st["l"] + "0.bias"]))
            pl.call("eegclip_layernorm_silu_fwd", _p(b[f"lin{s}"]), _p(P[st["l"] + "1.weight"]), _p(P[st["l"] + "1.bias"]), _p(b[f"ln{s}"]), _p(b[f"act{s}"]),
                    _p(b[f"mu{s}"]), _p(b[f"rs{s}"]), N, ho, 1e-5, p, 0, s, seed_at=11)
            if st["dec"] is not None:
                pl.call("eegclip_axpby", _p(b[skips[n_enc - 1 - st["dec"]]]), _p(b[f"act{s}"]), N * ho, 1.0, 1.0)     # x += hidden_activations[-1-j]
            cur = f"act{s}"
        pl.gemm(N, E, h0, _p(b[cur]), D(h0), D(1), _p(P["output_layer.weight"]), D(1), D(h0), _p(b["out"]), D(E), D(1), bias_n=_p(P["output_layer.bias"]))
        return pl

    # ---- sampling chain (Pipe.generate): chain-invariant embeddings hoisted, 20 launches per DDPM step ---------------------------------
    def sampling(self, N, S, cond_rows):
        """Plans + buffers for a sampling chain of S steps over N rows (2 x embeddings under classifier-free guidance), the first cond_rows of
        them conditioned.  In Pipe.generate every row shares the timestep and the condition is fixed, so
          * hoist plan (once per chain): time embeddings of all S timesteps for all 8 stages, TE_s (S, h_s), and the condition embeddings
            CE_s (cond_rows, h_s)  -- diffusion_prior.py:188-189,196-197 for every t of the schedule at once;
          * step plan (per DDPM step j): input Linear, then per stage ONE skinny GEMM + ONE eegclip_prior_stage_infer (LayerNorm, SiLU, skip
            add, + TE_s[j] + CE_s for the next stage), output Linear.
        The reference evaluates 2 x 34 Linear layers + 18 small elementwise ops per step (diffusion_prior.py:362-367)."""
        key = ("s", N, S, cond_rows)
        if key in self.plans:
            return self.plans[key]
        P, m, dev = self.P, self.model, self.device
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        E, Td, Cd, h0 = m.embed_dim, m.time_embed_dim, m.cond_dim, m.hidden_dim[0]
        b = dict(ts=f(S), temb=f(S, Td), x=f(N, E), linI=f(N, h0), actI=f(N, h0), out=f(N, E), c=f(max(cond_rows, 1), Cd))
        for s, st in enumerate(self.stages):
            hi, ho = st["hin"], st["hout"]
            b.update({f"t1{s}": f(S, hi), f"te{s}": f(S, hi), f"ce{s}": f(max(cond_rows, 1), hi), f"xin{s}": f(N, hi), f"lin{s}": f(N, ho), f"act{s}": f(N, ho)})
        hoist = Plan(f"prior_hoist[S={S}]")
        hoist.call("eegclip_timestep_embedding", _p(b["ts"]), S, Td, _p(b["temb"]))
        for s, st in enumerate(self.stages):
            hi = st["hin"]
            hoist.gemm(S, hi, Td, _p(b["temb"]), D(Td), D(1), _p(P[st["t"] + "linear_1.weight"]), D(1), D(Td), _p(b[f"t1{s}"]), D(hi), D(1),
                       bias_n=_p(P[st["t"] + "linear_1.bias"]), act=ACT_SILU)
            hoist.gemm(S, hi, hi, _p(b[f"t1{s}"]), D(hi), D(1), _p(P[st["t"] + "linear_2.weight"]), D(1), D(hi), _p(b[f"te{s}"]), D(hi), D(1),
                       bias_n=_p(P[st["t"] + "linear_2.bias"]))
            if cond_rows:
                hoist.gemm(cond_rows, hi, Cd, _p(b["c"]), D(Cd), D(1), _p(P[st["c"] + "weight"]), D(1), D(Cd), _p(b[f"ce{s}"]), D(hi), D(1),
                           bias_n=_p(P[st["c"] + "bias"]))
        step = Plan(f"prior_step[N={N}]")
        step.te_slots = []                   # (op index, stage, width): argument 5 of these ops is TE_stage + j * width floats
        n_st, n_enc = len(self.stages), m.num_layers - 1

        def finish(lin, gamma, beta, skip, act, nxt, rows_h):
            idx = len(step.ops)
            if nxt is None:
                step.call("eegclip_prior_stage_infer", lin, gamma, beta, skip, act, None, None, 0, None, N, rows_h, 1e-5)
            else:
                step.call("eegclip_prior_stage_infer", lin, gamma, beta, skip, act, _p(b[f"te{nxt}"]), _p(b[f"ce{nxt}"]) if cond_rows else None,
                          cond_rows, _p(b[f"xin{nxt}"]), N, rows_h, 1e-5)
                step.te_slots.append((idx, nxt, self.stages[nxt]["hin"]))

        step.gemm(N, h0, E, _p(b["x"]), D(E), D(1), _p(P["input_layer.0.weight"]), D(1), D(E), _p(b["linI"]), D(h0), D(1), bias_n=_p(P["input_layer.0.bias"]))
        finish(_p(b["linI"]), _p(P["input_layer.1.weight"]), _p(P["input_layer.1.bias"]), None, _p(b["actI"]), 0, h0)
        cur, skips = "actI", []
        for s, st in enumerate(self.stages):
            hi, ho = st["hin"], st["hout"]
            if st["dec"] is None:
                skips.append(cur)
            step.gemm(N, ho, hi, _p(b[f"xin{s}"]), D(hi), D(1), _p(P[st["l"] + "0.weight"]), D(1), D(hi), _p(b[f"lin{s}"]), D(ho), D(1),
                      bias_n=_p(P[st["l"] + "0.bias"]))
            skip = _p(b[skips[n_enc - 1 - st["dec"]]]) if st["dec"] is not None else None          # x += hidden_activations[-1-j]
            finish(_p(b[f"lin{s}"]), _p(P[st["l"] + "1.weight"]), _p(P[st["l"] + "1.bias"]), skip, _p(b[f"act{s}"]), s + 1 if s + 1 < n_st else None, ho)
            cur = f"act{s}"
        step.gemm(N, E, h0, _p(b[cur]), D(h0), D(1), _p(P["output_layer.weight"]), D(1), D(h0), _p(b["out"]), D(E), D(1), bias_n=_p(P["output_layer.bias"]))
        self.plans[key] = (hoist, step, b)
        return self.plans[key]

    def _build_bwd(self, N, cond, p):
        P, G, b, m = self.P, self.G, self.bufs[N], self.model
        pl = Plan(f"prior_bwd[N={N}]")
        E, Td, Cd, h0 = m.embed_dim, m.time_embed_dim, m.cond_dim, m.hidden_dim[0]
        sk = lambda k: max(1, min(16, k // 256))

        def wgrad(name, dY, ldy, X, ldx, Nout, Nin, small, bias=None):
            # weight gradients are off the dX chain: they run on the plan's second stream underneath it (the prior's GEMMs are small -- a
            # 1024 x 128 gradient is 32 tiles on 256 CUs); the bias gradient is the row sum of dY^T the same launch already streams (rowsum_a)
            g = pl.gemm(Nout, Nin, N, dY, D(1), D(ldy), X, D(ldx), D(1), _p(G[name]), D(Nin), D(1), accumulate=1, split_k=sk(N) if small else 1,
                        rowsum_a=_p(G[bias]) if bias else None, side=True)
            return g

        n_st = len(self.stages)
        n_enc = m.num_layers - 1
        last = f"act{n_st - 1}"
        wgrad("output_layer.weight", _p(b["dout"]), E, _p(b[last]), h0, E, h0, False, bias="output_layer.bias")
        pl.gemm(N, h0, E, _p(b["dout"]), D(E), D(1), _p(P["output_layer.weight"]), D(h0), D(1), _p(b[f"dact{n_st - 1}"]), D(h0), D(1))
        pl.c_gemms = []
        for s in range(n_st - 1, -1, -1):
            st = self.stages[s]
            hi, ho = st["hin"], st["hout"]
            small = hi * ho < 256 * 256
            pl.call("eegclip_silu_bwd", _p(b[f"dact{s}"]), _p(b[f"ln{s}"]), _p(b[f"dln{s}"]), N * ho, 0, p, 0, s, seed_at=6)
            pl.call("eegclip_layernorm_bwd", _p(b[f"dln{s}"]), _p(b[f"lin{s}"]), _p(P[st["l"] + "1.weight"]), _p(b[f"mu{s}"]), _p(b[f"rs{s}"]), _p(b[f"dlin{s}"]),
                    _p(G[st["l"] + "1.weight"]), _p(G[st["l"] + "1.bias"]), N, ho, 0, None, 0.0, 0, 0)
            wgrad(st["l"] + "0.weight", _p(b[f"dlin{s}"]), ho, _p(b["XIN"]) + 4 * self.col0[s], self.wide, ho, hi, small, bias=st["l"] + "0.bias")
            # dxin = dlin W (kept pure in Cpre: the weight-gradient GEMMs on the second stream read it) and, in the same epilogue, the gradient
            # w.r.t. the stage input: dst = dxin (+ the skip branch for encoder stages: decode stage j = n_enc-1-i adds skips[i]) -- two
            # elementwise launches per stage fewer
            dst = _p(b[f"dact{s - 1}"]) if s > 0 else _p(b["dactI"])
            skip = _p(b[f"dact{n_enc + (n_enc - 1 - s)}"]) if st["dec"] is None else None
            pl.gemm(N, hi, ho, _p(b[f"dlin{s}"]), D(ho), D(1), _p(P[st["l"] + "0.weight"]), D(hi), D(1), dst, D(hi), D(1), Cpre=_p(b[f"dxin{s}"]),
                    R=skip, Rm=D(hi) if skip else D(0), Rn=D(1) if skip else D(0))
            if cond:
                pl.c_gemms.append(wgrad(st["c"] + "weight", _p(b[f"dxin{s}"]), hi, 0, Cd, hi, Cd, hi < 256, bias=st["c"] + "bias"))
            wgrad(st["t"] + "linear_2.weight", _p(b[f"dxin{s}"]), hi, _p(b["T1act"]) + 4 * self.col0[s], self.wide, hi, hi, hi < 256,
                  bias=st["t"] + "linear_2.bias")
            # gradient of the time embedding's hidden layer, into this stage's columns of DT1 (SiLU' and the weight gradient of all eight
            # first Linears follow once, after the loop)
            pl.gemm(N, hi, hi, _p(b[f"dxin{s}"]), D(hi), D(1), _p(P[st["t"] + "linear_2.weight"]), D(hi), D(1), _p(b["DT1"]) + 4 * self.col0[s],
                    D(self.wide), D(1))
        pl.call("eegclip_silu_bwd", _p(b["DT1"]), _p(b["T1pre"]), _p(b["DT1"]), N * self.wide, 0, 0.0, 0, 0)
        wgrad(self.stages[0]["t"] + "linear_1.weight", _p(b["DT1"]), self.wide, _p(b["temb"]), Td, self.wide, Td, False,
              bias=self.stages[0]["t"] + "linear_1.bias")                       # the grouped (wide x 512) block of all eight stages
        pl.call("eegclip_silu_bwd", _p(b["dactI"]), _p(b["lnI"]), _p(b["dlnI"]), N * h0, 0, 0.0, 0, 0)
        pl.call("eegclip_layernorm_bwd", _p(b["dlnI"]), _p(b["linI"]), _p(P["input_layer.1.weight"]), _p(b["muI"]), _p(b["rsI"]), _p(b["dlinI"]),
                _p(G["input_layer.1.weight"]), _p(G["input_layer.1.bias"]), N, h0, 0, None, 0.0, 0, 0)
        pl.x_gemm = wgrad("input_layer.0.weight", _p(b["dlinI"]), h0, 0, E, h0, E, False, bias="input_layer.0.bias")
        return pl

    # ---- training plans over bf16 hi | lo PLANES (round 4) ---------------------------------------------------------------------------------
    # Every Linear of the step takes its operands as planes and leaves its result as planes for the next one (csrc/gemm_planes.hip: C = A B^T;
    # csrc/wgrad_tok.hip: the weight gradients, contraction over the batch through LDS transpose reads), so nothing is converted inside a GEMM.
    # The general split-bf16 GEMM split both fp32 operands in every workgroup of every launch: ~20 us per launch on these shapes whatever their
    # size, 37 us for each of the 27 weight gradients (both operands k-strided) -- 64 launches, 1.33 of the step's 1.5 ms.
    def _planes_ok(self, N, cond_rows):
        m = self.model
        dims = [m.embed_dim, m.time_embed_dim] + list(m.hidden_dim) + ([m.cond_dim] if m.cond_dim else [])
        # (eegclip_split_transpose takes at most 24 items -- GPT_MAX of csrc/gemm_planes.hip -- and the planes plan passes 2 per stage + 1: deeper stacks keep the
        #  general GEMM plans)
        return (cond_rows is None and N % 64 == 0 and all(d % 64 == 0 for d in dims) and default_gemm_precision() == _abi.PREC_BF16X3
                and 2 * len(self.stages) + 1 <= 24 and os.environ.get("EEGCLIP_PRIOR_PLANES", "1") != "0")

    def _alloc_planes(self, N, b):
        if "xp" in b:
            return
        dev, m, W = self.device, self.model, self.wide
        bf = lambda rows, cols: torch.zeros(2, rows * cols + 256, dtype=torch.bfloat16, device=dev)      # [hi | lo]; 512 bytes of slack: the weight-gradient
        #                                                                                                    kernel reads whole 128-channel tiles
        b.update(xp=bf(N, m.embed_dim), cp=bf(N, m.cond_dim), tembp=bf(N, m.time_embed_dim), T1actP=bf(N, W), XINP=bf(N, W), actLP=bf(N, m.hidden_dim[0]),
                 doutP=bf(N, m.embed_dim), DXINP=bf(N, W), DT1P=bf(N, W), dlinIP=bf(N, m.hidden_dim[0]))
        for s, st in enumerate(self.stages):
            b[f"dlinP{s}"] = bf(N, st["hout"])
        if not hasattr(self, "wp"):
            self.wp = torch.zeros(2, self.flat.numel(), dtype=torch.bfloat16, device=dev)       # planes of the whole flat parameter buffer, same offsets
            self.wt = {}                                                                        # transposed planes of the weights the dX GEMMs contract over
            for st in self.stages:
                for k in (st["l"] + "0.weight", st["t"] + "linear_2.weight"):
                    r, c = self.P[k].shape
                    self.wt[k] = torch.zeros(2, c * r, dtype=torch.bfloat16, device=dev)
            r, c = self.P["output_layer.weight"].shape
            self.wt["output_layer.weight"] = torch.zeros(2, c * r, dtype=torch.bfloat16, device=dev)

    def _wplanes(self, k):
        off = (self.P[k].data_ptr() - self.flat.data_ptr()) // 2            # bytes into a bf16 plane = half the fp32 byte offset
        return self.wp[0].data_ptr() + off, self.wp[1].data_ptr() + off

    @staticmethod
    def _pl(t, col=0):
        return t[0].data_ptr() + 2 * col, t[1].data_ptr() + 2 * col

    def _gp(self, pl, A, lda, Bw, ldb, M, Nn, K, side=False, **kw):
        a_hi, a_lo = A
        b_hi, b_lo = Bw
        d = _abi.GemmPlanesDesc(a_hi=a_hi, a_lo=a_lo, b_hi=b_hi, b_lo=b_lo, lda=lda, ldb=ldb, M=M, N=Nn, K=K, **kw)
        return pl.call_desc("eegclip_gemm_planes", d, side=side)

    def _build_fwd_planes(self, N, cond, p):
        P, b, m = self.P, self.bufs[N], self.model
        self._alloc_planes(N, b)
        pl = Plan(f"prior_fwd_planes[N={N}]")
        pl.planes = True
        E, Td, Cd, h0, W = m.embed_dim, m.time_embed_dim, m.cond_dim, m.hidden_dim[0], self.wide
        pl.call("eegclip_timestep_embedding", _p(b["tt"]), N, Td, _p(b["temb"]))
        # this step's weights as planes: the flat parameter buffer in one pass (a Linear's planes sit at its offset), the transposes the dX GEMMs need
        pl.call("eegclip_split_bf16", _p(self.flat), self.wp[0].data_ptr(), self.wp[1].data_ptr(), self.flat.numel())
        items = []
        for k, t in self.wt.items():
            r, c = P[k].shape
            items.append(_abi.SplitItem(src=_p(P[k]), hi=t[0].data_ptr(), lo=t[1].data_ptr(), rows=r, cols=c, ld_src=c, ld_out=r, transpose=1))
        arr = (_abi.SplitItem * len(items))(*items)
        pl._keep.append(arr)
        pl.call("eegclip_split_transpose", arr, len(items))
        # the inputs: x, c (patched per call) and the timestep sinusoid
        ins = [_abi.SplitItem(src=0, hi=b["xp"][0].data_ptr(), lo=b["xp"][1].data_ptr(), rows=N, cols=E, ld_src=E, ld_out=E, transpose=0),
               _abi.SplitItem(src=_p(b["temb"]), hi=b["tembp"][0].data_ptr(), lo=b["tembp"][1].data_ptr(), rows=N, cols=Td, ld_src=Td, ld_out=Td, transpose=0)]
        if cond:
            ins.append(_abi.SplitItem(src=0, hi=b["cp"][0].data_ptr(), lo=b["cp"][1].data_ptr(), rows=N, cols=Cd, ld_src=Cd, ld_out=Cd, transpose=0))
        pl.in_items = (_abi.SplitItem * len(ins))(*ins)
        pl._keep.append(pl.in_items)
        pl.call("eegclip_split_rows", pl.in_items, len(ins))
        self._gp(pl, self._pl(b["xp"]), E, self._wplanes("input_layer.0.weight"), E, N, h0, E, C=_p(b["linI"]), ldc=h0, bias=_p(P["input_layer.0.bias"]))
        pl.call("eegclip_prior_stage_fwd", _p(b["linI"]), _p(P["input_layer.1.weight"]), _p(P["input_layer.1.bias"]), None, _p(b["lnI"]), _p(b["actI"]),
                _p(b["muI"]), _p(b["rsI"]), None, None, N, h0, 1e-5, 0.0, 0, 0)
        s0 = self.stages[0]
        # all eight time-embedding hidden layers: SiLU(t_emb W1^T + b1) as planes (pre-activation kept in fp32 for the backward)
        self._gp(pl, self._pl(b["tembp"]), Td, self._wplanes(s0["t"] + "linear_1.weight"), Td, N, W, Td, Cpre=_p(b["T1pre"]), ldcpre=W,
                 bias=_p(P[s0["t"] + "linear_1.bias"]), act=ACT_SILU, p_hi=b["T1actP"][0].data_ptr(), p_lo=b["T1actP"][1].data_ptr(), ldp=W, planes_of=1)
        if cond:
            self._gp(pl, self._pl(b["cp"]), Cd, self._wplanes(s0["c"] + "weight"), Cd, N, W, Cd, C=_p(b["XIN"]), ldc=W, bias=_p(P[s0["c"] + "bias"]))
        cur, skips = "actI", []
        n_enc, n_st = m.num_layers - 1, len(self.stages)
        for s, st in enumerate(self.stages):
            hi, ho, c0 = st["hin"], st["hout"], self.col0[s]
            if st["dec"] is None:
                skips.append(cur)
            xh, xl = self._pl(b["XINP"], c0)
            # stage input x + t_emb (+ c_emb): fp32 accumulator column block of XIN, and its planes for the stage Linear and its weight gradient
            self._gp(pl, self._pl(b["T1actP"], c0), W, self._wplanes(st["t"] + "linear_2.weight"), hi, N, hi, hi, C=_p(b["XIN"]) + 4 * c0, ldc=W,
                     bias=_p(P[st["t"] + "linear_2.bias"]), R=_p(b[cur]), ldr=hi, accumulate=1 if cond else 0, p_hi=xh, p_lo=xl, ldp=W, planes_of=1)
            self._gp(pl, (xh, xl), W, self._wplanes(st["l"] + "0.weight"), hi, N, ho, hi, C=_p(b[f"lin{s}"]), ldc=ho, bias=_p(P[st["l"] + "0.bias"]))
            last = s == n_st - 1
            skip = _p(b[skips[n_enc - 1 - st["dec"]]]) if st["dec"] is not None else None          # x += hidden_activations[-1-j]
            pl.call("eegclip_prior_stage_fwd", _p(b[f"lin{s}"]), _p(P[st["l"] + "1.weight"]), _p(P[st["l"] + "1.bias"]), skip, _p(b[f"ln{s}"]), _p(b[f"act{s}"]),
                    _p(b[f"mu{s}"]), _p(b[f"rs{s}"]), b["actLP"][0].data_ptr() if last else None, b["actLP"][1].data_ptr() if last else None, N, ho, 1e-5, p, 0, s,
                    seed_at=14)
            cur = f"act{s}"
        self._gp(pl, self._pl(b["actLP"]), h0, self._wplanes("output_layer.weight"), h0, N, E, h0, C=_p(b["out"]), ldc=E, bias=_p(P["output_layer.bias"]))
        return pl

    def _build_bwd_planes(self, N, cond, p):
        P, G, b, m = self.P, self.G, self.bufs[N], self.model
        pl = Plan(f"prior_bwd_planes[N={N}]")
        pl.planes = True
        E, Td, Cd, h0, W = m.embed_dim, m.time_embed_dim, m.cond_dim, m.hidden_dim[0], self.wide

        def wgrad(problems):
            """weight (+ bias) gradients on the plan's second stream: (weight key, bias key, dY planes, ld, M, X planes, ld, N); small outputs are
            K-sliced (fp32 atomics onto few addresses), large ones are one read-modify-write pass"""
            arr = (_abi.WgradPlanesProblem * len(problems))()
            for i, (wk, bk, dy, lda, M, x, ldb, Nn) in enumerate(problems):
                tiles = ((M + 127) // 128) * ((Nn + 127) // 128)
                slices = 1 if tiles >= 48 else max(1, min(8, N // 32 // 4, 64 // tiles))
                arr[i] = _abi.WgradPlanesProblem(a_hi=dy[0], a_lo=dy[1], lda=lda, b_hi=x[0], b_lo=x[1], ldb=ldb, rows=N, M=M, N=Nn, out=_p(G[wk]), ldo=Nn,
                                                 bias_out=_p(G[bk]), slices=slices)
            pl._keep.append(arr)
            pl.call("eegclip_wgrad_planes", arr, len(problems), side=not merge)

        # EEGCLIP_PRIOR_WGRAD_MERGE (default 1): the 22 weight gradients wait for the end of the dX chain and run as two launches of <= 12 problems instead
        # of eleven launches "beside" the chain -- a weight-gradient workgroup (128 KB of LDS) and a plane-GEMM workgroup (64 KB x 2) exclude each other on
        # a CU, so the second stream did not overlap, it queued (main 0.70 ms + second stream 0.33 ms = the 1.03 ms span of the trace).  Every operand
        # (per-stage dY / X planes) is still intact at the end of the backward.
        merge = os.environ.get("EEGCLIP_PRIOR_WGRAD_MERGE", "1") != "0"
        pending = []
        if merge:
            wgrad_now = wgrad
            wgrad = lambda problems: pending.extend(problems)       # noqa: E731

        n_st, n_enc = len(self.stages), m.num_layers - 1
        pl.call("eegclip_split_bf16", _p(b["dout"]), b["doutP"][0].data_ptr(), b["doutP"][1].data_ptr(), N * E)
        wt = lambda k: (self.wt[k][0].data_ptr(), self.wt[k][1].data_ptr())
        wgrad([("output_layer.weight", "output_layer.bias", self._pl(b["doutP"]), E, E, self._pl(b["actLP"]), h0, h0)])
        self._gp(pl, self._pl(b["doutP"]), E, wt("output_layer.weight"), E, N, h0, E, C=_p(b[f"dact{n_st - 1}"]), ldc=h0)
        for s in range(n_st - 1, -1, -1):
            st = self.stages[s]
            hi, ho, c0 = st["hin"], st["hout"], self.col0[s]
            dl = self._pl(b[f"dlinP{s}"])
            ws = b.setdefault(f"psb_ws{s}", torch.empty(int(lib().eegclip_prior_stage_bwd_workspace_floats(N, ho)), dtype=torch.float32, device=self.device))
            pl.call("eegclip_prior_stage_bwd", _p(b[f"dact{s}"]), _p(b[f"ln{s}"]), _p(b[f"lin{s}"]), _p(P[st["l"] + "1.weight"]), _p(b[f"mu{s}"]), _p(b[f"rs{s}"]),
                    None, dl[0], dl[1], None, None, N, ho, p, 0, s, _p(ws), seed_at=14)
            pl.call("eegclip_prior_stage_bwd_params", _p(ws), N, ho, _p(G[st["l"] + "1.weight"]), _p(G[st["l"] + "1.bias"]), side=True)
            # dxin = dlin W: as planes (operand of the time-embedding dX GEMM and of three weight gradients); in the same epilogue the gradient w.r.t.
            # the previous activation = dxin (+ the skip branch for encoder stages: decode stage j = n_enc-1-i adds skips[i])
            dst = _p(b[f"dact{s - 1}"]) if s > 0 else _p(b["dactI"])
            skip = _p(b[f"dact{n_enc + (n_enc - 1 - s)}"]) if st["dec"] is None else None
            dx = self._pl(b["DXINP"], c0)
            self._gp(pl, dl, ho, wt(st["l"] + "0.weight"), ho, N, hi, ho, C=dst, ldc=hi, R=skip, ldr=hi, p_hi=dx[0], p_lo=dx[1], ldp=W, planes_of=2)
            wgrad([(st["l"] + "0.weight", st["l"] + "0.bias", dl, ho, ho, self._pl(b["XINP"], c0), W, hi),
                   (st["t"] + "linear_2.weight", st["t"] + "linear_2.bias", dx, W, hi, self._pl(b["T1actP"], c0), W, hi)])
            # gradient of the time embedding's hidden layer, into this stage's columns of DT1
            self._gp(pl, dx, W, wt(st["t"] + "linear_2.weight"), hi, N, hi, hi, C=_p(b["DT1"]) + 4 * c0, ldc=W)
        s0 = self.stages[0]
        if cond:                                   # all eight condition Linears: ONE weight gradient (wide x cond_dim) from the wide dxin planes
            wgrad([(s0["c"] + "weight", s0["c"] + "bias", self._pl(b["DXINP"]), W, W, self._pl(b["cp"]), Cd, Cd)])
        pl.call("eegclip_silu_bwd_planes", _p(b["DT1"]), _p(b["T1pre"]), b["DT1P"][0].data_ptr(), b["DT1P"][1].data_ptr(), N * W)
        di = self._pl(b["dlinIP"])
        ws = b.setdefault("psb_wsI", torch.empty(int(lib().eegclip_prior_stage_bwd_workspace_floats(N, h0)), dtype=torch.float32, device=self.device))
        pl.call("eegclip_prior_stage_bwd", _p(b["dactI"]), _p(b["lnI"]), _p(b["linI"]), _p(P["input_layer.1.weight"]), _p(b["muI"]), _p(b["rsI"]), None, di[0], di[1],
                None, None, N, h0, 0.0, 0, 0, _p(ws))
        pl.call("eegclip_prior_stage_bwd_params", _p(ws), N, h0, _p(G["input_layer.1.weight"]), _p(G["input_layer.1.bias"]), side=True)
        # the last two weight gradients (the eight first time-embedding Linears as one wide x 512 block, the input Linear) are ONE launch: nothing is
        # left to run under them, and together their 92 + 64 output tiles are one wave of workgroups
        wgrad([(s0["t"] + "linear_1.weight", s0["t"] + "linear_1.bias", self._pl(b["DT1P"]), W, W, self._pl(b["tembp"]), Td, Td),
               ("input_layer.0.weight", "input_layer.0.bias", di, h0, h0, self._pl(b["xp"]), E, E)])
        # (eegclip_wgrad_planes: <= 24 problems per launch -- ONE launch since round 6: two launches of 360 + 444 workgroups were 2 + 2 rounds on 256 CUs,
        #  804 workgroups are 3.1; EEGCLIP_PRIOR_WGRAD_CHUNK=12: the two launches, A/B aid)
        chunk = int(os.environ.get("EEGCLIP_PRIOR_WGRAD_CHUNK", "24"))
        for i in range(0, len(pending), chunk):
            wgrad_now(pending[i:i + chunk])
        return pl

    def forward(self, x, t, c, p, cond_rows=None):
        """cond_rows (inference only): the condition embeddings `c` (cond_rows, cond_dim) apply to the FIRST cond_rows rows of x, the rest run
        unconditioned -- a classifier-free-guidance pair in one pass of 2N rows (the condition term is a separate accumulate-GEMM,
        so it simply covers fewer rows)"""
        N = x.shape[0]
        if N not in self.bufs:
            self.bufs[N] = self._alloc(N)
        b = self.bufs[N]
        cond = c is not None
        key = (N, cond, p) if cond_rows is None else (N, cond, p, cond_rows)
        pk = ("f",) + key
        if pk not in self.plans:
            self.plans[pk] = self._build_fwd_planes(N, cond, p) if self._planes_ok(N, cond_rows) else self._build_fwd(N, cond, p, cond_rows)
        pl = self.plans[pk]
        b["tt"].copy_(t)
        if getattr(pl, "planes", False):
            pl.in_items[0].src = x.data_ptr()
            if cond:
                pl.in_items[2].src = c.data_ptr()
        else:
            pl.x_gemm.A = x.data_ptr()
            for g in pl.c_gemms:
                g.A = c.data_ptr()
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        b["seed"] = seed
        pl.run(raw_stream(), seed)
        self.last_key = key
        self.version[key] = self.version.get(key, 0) + 1
        return b["out"]

    def attach_grads(self, cond):
        live = [k for k in self.params if cond or k not in self.cond_keys]
        mine = lambda k: self.params[k].grad is not None and self.params[k].grad.data_ptr() == self.G[k].data_ptr()
        if all(mine(k) for k in live):
            return
        if all(self.params[k].grad is None for k in self.params):
            self.gflat.zero_()
        else:
            for k in live:
                if not mine(k):
                    self.G[k].zero_()
        for k in live:
            self.params[k].grad = self.G[k]

    def backward(self, key, x, c, dout):
        N, cond, p = key
        b = self.bufs[N]
        pk = ("b",) + key
        if pk not in self.plans:
            fwd = self.plans.get(("f",) + key)
            self.plans[pk] = self._build_bwd_planes(N, cond, p) if getattr(fwd, "planes", False) else self._build_bwd(N, cond, p)
        pl = self.plans[pk]
        self.attach_grads(cond)
        if dout.data_ptr() != b["dout"].data_ptr():
            b["dout"].copy_(dout)
        if not getattr(pl, "planes", False):
            pl.x_gemm.B = x.data_ptr()
            for g in pl.c_gemms:
                g.B = c.data_ptr()
        pl.run(raw_stream(), b.get("seed", 0))


class _TrainStepPlan:
    """Pipe.train's steady-state iteration (Generation/diffusion_prior.py:302-333: add_noise, forward, MSE, backward, clip_grad_norm_, optimizer.step) as ONE
    launch plan replayed by one foreign call -- what step_plan.StepPlan is for the contrastive loop.  Built from the engine's own forward / backward plans of a
    (batch size, condition present, dropout) key and the optimizer's cached launches once the ordinary path has run that key; per step the host draws the
    noise and the timesteps into fixed buffers (the same generator calls in the same order), patches four pointers, the learning rate and the step counts.
    Launch for launch the ordinary step, except that the optimizer launches clear the gradients behind their read (the zero_grad() that opens the next
    iteration) instead of a 39 MB fill.  Single process only (data parallel: the gradient all-reduce sits between backward and clipping)."""

    def __init__(self, pipe, eng, optimizer, key, fast, sumsq, clip):
        N, cond, p = key
        self.key, self.fast, self.eng = key, fast, eng
        fwd, bwd = eng.plans[("f",) + key], eng.plans[("b",) + key]
        if not (getattr(fwd, "planes", False) and getattr(bwd, "planes", False)):
            raise ValueError("plane-GEMM plans required")
        b = eng.bufs[N]
        E, dev = eng.model.embed_dim, eng.device
        T = pipe.scheduler.config.num_train_timesteps
        self.T = T
        self.noise = torch.empty(N, E, dtype=torch.float32, device=dev)
        self.ts = torch.empty(N, dtype=torch.int64, device=dev)
        self.x = torch.empty(N, E, dtype=torch.float32, device=dev)
        sa, sb = pipe.scheduler._tables(dev)
        pl = Plan(f"prior_train_step[N={N}]")
        pl._keep += [fwd, bwd, sa, sb, self.noise, self.ts, self.x, sumsq, clip]

        def splice(src):
            base = len(pl.ops)
            for fn, args, name, side in src.ops:
                pl.ops.append((fn, list(args), name, side))
            pl._seed_slots += [(base + i, j) for i, j in src._seed_slots]
            pl._seed_descs += src._seed_descs
        self.noise_op = len(pl.ops)
        pl.call("eegclip_ddpm_add_noise", 0, _p(self.noise), _p(self.ts), _p(sa), _p(sb), _p(self.x), N, E)
        self.fwd = fwd
        splice(fwd)
        self.mse_op = len(pl.ops)
        pl.call("eegclip_mse_loss_grad", _p(b["out"]), _p(self.noise), N * E, 0, _p(b["dout"]))
        splice(bwd)
        pl.join()
        pl.memset(sumsq)
        pl.call("eegclip_sumsq", _p(eng.gflat), eng.gflat.numel(), _p(sumsq))
        pl.call("eegclip_clip_scale", _p(sumsq), 1.0, _p(clip))
        g = optimizer.param_groups[0]
        b1, b2 = g["betas"]
        self.adam_ops = []
        for (p0, n, wp, gp, mp, vp, members) in fast["launch"]:
            self.adam_ops.append(len(pl.ops))
            pl.call("eegclip_adamw_step_zero_grad", wp, gp, mp, vp, n, g["lr"], b1, b2, g["eps"], g["weight_decay"], 0, 1.0, _p(clip))
        self.pl = pl
        self.group = g

    def usable(self, optimizer):
        eng, key = self.eng, self.key
        return eng.plans.get(("f",) + key) is self.fwd and optimizer.param_groups[0] is self.group and optimizer.activate_launch_set(0, self.fast)

    def run(self, h, c, loss_sum, tt):
        """h (N, E) fp32 contiguous on the device, c the condition or None (as the key says); draws noise / timesteps like the ordinary step"""
        pl, fast, g = self.pl, self.fast, self.group
        N, cond, p = self.key
        self.noise.normal_()                                   # torch.randn_like(h): the same generator call
        self.ts.random_(0, self.T)                             # torch.randint(0, T, (N,), device=...)
        tt.copy_(self.ts)                                      # the time embedding's fp32 timesteps
        pl.set_arg(self.noise_op, 0, h.data_ptr())
        pl.set_arg(self.mse_op, 3, loss_sum.data_ptr())
        self.fwd.in_items[0].src = self.x.data_ptr()
        if cond:
            self.fwd.in_items[2].src = c.data_ptr()
        fast["run_steps"] = [st + 1 for st in fast["run_steps"]]
        fast["pending"] += 1
        for op, st in zip(self.adam_ops, fast["run_steps"]):
            pl.set_arg(op, 5, g["lr"])
            pl.set_arg(op, 10, st)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        self.eng.bufs[N]["seed"] = seed
        pl._keep_step = (h, c, loss_sum)
        pl.run(raw_stream(), seed)
        self.eng.last_key = self.key
        self.eng.version[self.key] = self.eng.version.get(self.key, 0) + 1


class EmbeddingDataset(Dataset):
    def __init__(self, c_embeddings, h_embeddings):
        self.c_embeddings = c_embeddings
        self.h_embeddings = h_embeddings

    def __len__(self):
        return len(self.c_embeddings)

    def __getitem__(self, idx):
        return {"c_embedding": self.c_embeddings[idx], "h_embedding": self.h_embeddings[idx]}


class DDPMScheduler:
    """diffusers-0.30.0 DDPMScheduler defaults, restated (parity unpinned, see oracle/prior.py): 1000 steps, linear betas
    1e-4..0.02, epsilon prediction, variance fixed_small, clip_sample to [-1,1], "leading" timestep spacing."""

    class _Cfg:
        num_train_timesteps = 1000
        prediction_type = "epsilon"
        clip_sample = True
        clip_sample_range = 1.0

    class _Out:
        def __init__(self, prev):
            self.prev_sample = prev

    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        self.config = self._Cfg()
        betas = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self._acp = [float(v) for v in self.alphas_cumprod]
        self.num_inference_steps = None
        self.timesteps = torch.arange(999, -1, -1)
        self._tabs = {}

    def _tables(self, device):
        if device not in self._tabs:
            self._tabs[device] = (self.alphas_cumprod.sqrt().to(device).contiguous(), (1 - self.alphas_cumprod).sqrt().to(device).contiguous())
        return self._tabs[device]

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        self.timesteps = (torch.arange(0, num_inference_steps) * ratio).flip(0).long()      # kept on the HOST: no .item() sync per step

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        require_cuda(original_samples, "original_samples")
        sa, sb = self._tables(original_samples.device)
        h, nz = original_samples.float().contiguous(), noise.float().contiguous()
        out = torch.empty_like(h)
        n = h.shape[0]
        check(lib().eegclip_ddpm_add_noise(h.data_ptr(), nz.data_ptr(), timesteps.to(h.device).long().contiguous().data_ptr(), sa.data_ptr(),
                                           sb.data_ptr(), out.data_ptr(), n, h.numel() // n, raw_stream()), "ddpm_add_noise")
        return out

    def step_coeffs(self, t):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        tp = t - self.config.num_train_timesteps // n
        acp_t = self._acp[t]
        acp_prev = self._acp[tp] if tp >= 0 else 1.0
        bt, bp = 1 - acp_t, 1 - acp_prev
        ca = acp_t / acp_prev
        cb = 1 - ca
        var = max(bp / bt * cb, 1e-20)
        return acp_t ** 0.5, bt ** 0.5, (acp_prev ** 0.5 * cb) / bt, ca ** 0.5 * bp / bt, (var ** 0.5 if t > 0 else 0.0)

    def draw_noise(self, shape, device, generator=None):
        """the variance noise of one step, drawn exactly like step() draws it (Pipe.generate pre-draws the whole chain in step order)"""
        gdev = generator.device if generator is not None else device
        return torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(device)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, model_output_uncond=None, guidance_scale=0.0, noise=None,
             out=None, out_dup=None):
        """x_t -> x_{t-1}.  `model_output_uncond` (+ guidance_scale) fuses the classifier-free-guidance mix into the same kernel.
        `noise`: the step's variance noise if the caller has drawn it already (ignored at t = 0, like the generator); `out` / `out_dup`:
        where to write x_{t-1} (out may be `sample` itself) and an optional second copy."""
        t = int(timestep)
        sa, sb, c0, ct, sigma = self.step_coeffs(t)
        x = sample.contiguous()
        if t == 0:
            noise = None
        elif noise is None:
            noise = self.draw_noise(x.shape, x.device, generator)
        out = torch.empty_like(x) if out is None else out
        eu = model_output_uncond
        check(lib().eegclip_ddpm_step(x.data_ptr(), model_output.contiguous().data_ptr(), eu.contiguous().data_ptr() if eu is not None else None,
                                      float(guidance_scale), sa, sb, c0, ct, sigma, noise.data_ptr() if noise is not None else None, out.data_ptr(),
                                      out_dup.data_ptr() if out_dup is not None else None, x.numel(), raw_stream()),
              "ddpm_step")
        return self._Out(out)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, **kw):
    scheduler.set_timesteps(num_inference_steps, device=device)
    return scheduler.timesteps, num_inference_steps


def cosine_with_warmup_lr(step, base_lr, warmup, total, cycles=0.5):
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * cycles * 2.0 * prog)))


class Pipe:
    def __init__(self, diffusion_prior=None, scheduler=None, device='cuda'):
        self.diffusion_prior = diffusion_prior.to(device)
        self.scheduler = scheduler if scheduler is not None else DDPMScheduler()
        self.device = device
        self.lr_history = []
        self.cond_drop_prob = 0.1            # whole-batch condition drop of the training loop (diffusion_prior.py:304)
        self.cond_dropped = []               # the decision taken at every update (diagnostics / tests)

    def train(self, dataloader, num_epochs=10, learning_rate=1e-4):
        from . import dist as edist
        from . import optim
        from .retrieval import settle_gc
        prior = self.diffusion_prior
        prior.train()
        settle_gc()                                   # no generation-2 collection (75 ms on this host) in the middle of an epoch
        device = self.device
        optimizer = optim.Adam(prior.parameters(), lr=learning_rate)
        total_steps = len(dataloader) * num_epochs
        T = self.scheduler.config.num_train_timesteps
        eng = prior._engine()
        L = lib()
        sumsq = torch.zeros(1, dtype=torch.float64, device=device)
        clip = torch.ones(1, dtype=torch.float32, device=device)
        optimizer.grad_scale_dev = clip
        step = 0
        drop_gen = None
        plans, warm = {}, {}
        plan_on = edist.world_size() == 1 and os.environ.get("EEGCLIP_PRIOR_STEP_PLAN", "1") != "0"
        if edist.world_size() > 1:
            # data parallel: the whole-batch condition drop must be ONE decision for all ranks -- a rank that dropped the condition has no
            # gradient for the condition layers, Adam would skip them there and step them elsewhere, and the replicas would drift apart for
            # good.  One seed from rank 0 (its global RNG stream, like the single-process draw), then every rank draws the same sequence.
            import torch.distributed as dist
            seed = torch.empty(1, dtype=torch.int64).random_(0, 2 ** 31 - 1)
            seed = seed.to(device) if dist.get_backend() == "nccl" else seed
            dist.broadcast(seed, 0)
            drop_gen = torch.Generator().manual_seed(int(seed))
        for epoch in range(num_epochs):
            loss_sum = torch.zeros(1, dtype=torch.float32, device=device)
            for batch in dataloader:
                c_embeds = batch['c_embedding'].to(device) if 'c_embedding' in batch.keys() else None
                h_embeds = batch['h_embedding'].to(device).float()
                N = h_embeds.shape[0]
                if (torch.rand(1, generator=drop_gen) if drop_gen is not None else torch.rand(1)) < self.cond_drop_prob:      # (:304)
                    c_embeds = None
                self.cond_dropped.append(c_embeds is None)
                step += 1
                lr = cosine_with_warmup_lr(step, learning_rate, 500, total_steps)      # lr_scheduler.step() BEFORE optimizer.step() (:331-332)
                self.lr_history.append(lr)
                for gq in optimizer.param_groups:
                    gq["lr"] = lr
                c32 = c_embeds.float().contiguous() if c_embeds is not None else None
                # the steady state as one submission (_TrainStepPlan): a (batch size, condition, dropout) key that the ordinary path below has stepped
                # twice -- its plans and the optimizer's launch set exist -- goes through its plan; EEGCLIP_PRIOR_STEP_PLAN=0: never
                tkey = (N, c32 is not None, prior.drop_p())
                tp = plans.get(tkey)
                if tp is not None and tp is not False and h_embeds.is_contiguous() and tp.usable(optimizer):
                    if any(q.grad is not None for q in optimizer.param_groups[0]["params"]):
                        optimizer.zero_grad()                           # (an ordinary step of another key came between: its gradients are spent, the
                        eng.gflat.zero_()                               #  plan accumulates into a clear buffer)
                    tp.run(h_embeds, c32, loss_sum, eng.bufs[N]["tt"])
                    continue
                if tp is not None and tp is not False:
                    plans.pop(tkey)                                     # something changed under the plan (the optimizer's runs were re-formed): warm up again
                    warm[tkey] = 0
                noise = torch.randn_like(h_embeds)
                timesteps = torch.randint(0, T, (N,), device=device)
                perturbed = self.scheduler.add_noise(h_embeds, noise, timesteps)
                st = raw_stream()
                optimizer.zero_grad()
                pred = eng.forward(perturbed, timesteps.float(), c32, prior.drop_p())
                b = eng.bufs[N]
                check(L.eegclip_mse_loss_grad(pred.data_ptr(), noise.data_ptr(), pred.numel(), loss_sum.data_ptr(), b["dout"].data_ptr(), st), "mse")
                eng.backward(eng.last_key, perturbed, c32, b["dout"])
                if edist.world_size() > 1:
                    edist.average_flat_grads(eng.gflat)
                sumsq.zero_()                                             # clip_grad_norm_(params, 1.0) without a host sync
                check(L.eegclip_sumsq(eng.gflat.data_ptr(), eng.gflat.numel(), sumsq.data_ptr(), st), "sumsq")
                check(L.eegclip_clip_scale(sumsq.data_ptr(), 1.0, clip.data_ptr(), st), "clip_scale")
                optimizer.step()
                warm[tkey] = warm.get(tkey, 0) + 1
                if plan_on and warm[tkey] >= 2 and tkey not in plans and optimizer._fast_last.get(0) is not None:
                    try:
                        plans[tkey] = _TrainStepPlan(self, eng, optimizer, tkey, optimizer._fast_last[0], sumsq, clip)
                        optimizer.zero_grad()                           # the plan accumulates into a CLEAR flat gradient buffer and leaves it clear
                        eng.gflat.zero_()
                    except (ValueError, KeyError):
                        plans[tkey] = False                             # (general-GEMM plans: launch by launch for good)
            loss_epoch = float(loss_sum) / len(dataloader)
            print(f'epoch: {epoch}, loss: {loss_epoch}')
        return self

    def generate(self, c_embeds=None, num_inference_steps=50, timesteps=None, guidance_scale=5.0, generator=None):
        prior = self.diffusion_prior
        prior.eval()
        N = c_embeds.shape[0] if c_embeds is not None else 1
        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, self.device, timesteps)
        if c_embeds is not None:
            c_embeds = c_embeds.to(self.device).float().contiguous()
        gdev = generator.device if generator is not None else self.device
        h_t = torch.randn(N, prior.embed_dim, generator=generator, device=gdev).to(self.device)
        steps = timesteps.tolist()                                         # host-side schedule: no device->host sync in the loop
        # the chain's variance noise, drawn up front in step order (the same generator calls, in the same order, as drawing inside step())
        noise = [self.scheduler.draw_noise(h_t.shape, h_t.device, generator) for t in steps if t > 0]
        noise = torch.stack(noise) if noise else torch.zeros(0, *h_t.shape, device=h_t.device)
        eng = prior._engine()
        use_cfg = not (guidance_scale == 0 or c_embeds is None)
        graphs = h_t.is_cuda and os.environ.get("EEGCLIP_PRIOR_GRAPH", "1") != "0"
        # the schedule's timesteps feed the hoisted time embeddings (read at run / replay time: uploaded before either)
        eng.sampling(2 * N if use_cfg else N, len(steps), N if use_cfg else 0)[2]["ts"].copy_(torch.tensor(steps, dtype=torch.float32))
        with torch.no_grad():
            if not graphs:
                return self._chain(eng, h_t, c_embeds if use_cfg else None, noise, steps, guidance_scale)
            # 20 small dependent launches per DDPM step: the WHOLE chain (hoisted embeddings + every step) is captured once into a HIP graph over
            # static buffers (start latent, condition, noise, timesteps) and replayed with one launch, so the host never paces the chain.
            # Weights are read through the flat parameter buffer at replay time: a trained / reloaded prior needs no re-capture.
            key = (N, tuple(steps), float(guidance_scale), use_cfg, id(eng), eng.flat.data_ptr())
            if not hasattr(self, "_graphs"):
                self._graphs = {}
            g = self._graphs.get(key)
            if g is None:
                if len(self._graphs) >= 8:
                    self._graphs.clear()
                st = dict(h0=torch.empty_like(h_t), noise=torch.empty_like(noise), c=torch.empty_like(c_embeds) if use_cfg else None)
                self._chain(eng, h_t, c_embeds if use_cfg else None, noise, steps, guidance_scale)      # launch by launch once: builds the plans
                torch.cuda.synchronize()
                st["graph"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(st["graph"]):
                    st["out"] = self._chain(eng, st["h0"], st["c"], st["noise"], steps, guidance_scale)
                g = self._graphs[key] = st
            g["h0"].copy_(h_t)
            g["noise"].copy_(noise)
            if use_cfg:
                g["c"].copy_(c_embeds)
            g["graph"].replay()
            return g["out"].clone()

    def _chain(self, eng, h_t, c_embeds, noise, steps, guidance_scale):
        """the DDPM ancestral chain (diffusion_prior.py:358-377) from latent h_t; c_embeds None = no classifier-free guidance.  Both
        predictions of a guided step are ONE pass over 2N rows (rows < N conditioned); see _PriorEngine.sampling for what is hoisted."""
        N, S = h_t.shape[0], len(steps)
        cfg = c_embeds is not None
        rows = 2 * N if cfg else N
        hoist, step, b = eng.sampling(rows, S, N if cfg else 0)
        stream = raw_stream()
        if cfg:
            b["c"].copy_(c_embeds)
        hoist.run(stream)
        x = b["x"]
        x[:N].copy_(h_t)
        if cfg:
            x[N:].copy_(h_t)
        j = 0
        for i, t in enumerate(steps):
            for idx, st, width in step.te_slots:
                step.set_arg(idx, 5, b[f"te{st}"].data_ptr() + 4 * width * i)
            step.run(stream)
            nz = None
            if t > 0:
                nz, j = noise[j], j + 1
            eps = b["out"]
            self.scheduler.step(eps[:N], t, x[:N], noise=nz, model_output_uncond=eps[N:] if cfg else None, guidance_scale=guidance_scale if cfg else 0.0,
                                out=x[:N], out_dup=x[N:] if cfg else None)
        return x[:N].clone()
