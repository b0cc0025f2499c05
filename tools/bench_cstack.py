"""stand-alone timings of the recomputing conv-stack kernels (csrc/cstack*.hip) at B = 256: event-bracketed loops of n launches (kernels >= 10 us)"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib

L = lib()
B, H, C, W = int(os.environ.get("CB_B", "256")), 63, 40, 36
st = torch.cuda.current_stream().cuda_stream
dev = "cuda"
x = torch.randn(B, 64, 250, device=dev)
w25, bias1 = torch.randn(40, 25, device=dev) * 0.2, torch.randn(40, device=dev) * 0.1
g1, b1 = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
Ws, bias2 = torch.randn(C, C, H, device=dev) / (C * H) ** 0.5, 0.1 * torch.randn(C, device=dev)
dy2 = torch.randn(B, C, W, device=dev)
packed = torch.empty(int(L.eegclip_cstack_packed_bytes(H)) // 2, dtype=torch.bfloat16, device=dev)
rows = torch.empty(2, B, 80, dtype=torch.float64, device=dev)
mu, rs = torch.empty(C, device=dev), torch.empty(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
nbt = torch.zeros(1, dtype=torch.long, device=dev)
y2, y1 = torch.empty(B, C, W, device=dev), torch.empty(B, C, H, W, device=dev)


def ev(f, n=20):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 2)


def chk(rc):
    assert rc == 0, rc


def fwd_desc(with_y1, train=True):
    return _abi.CstackFwdDesc(B=B, H=H, x=x.data_ptr(), xs_b=64 * 250, xs_h=250, w25=w25.data_ptr(), bias1=bias1.data_ptr(),
                              stat1=rows[0].data_ptr() if train else None, nstat1=B if train else 0, count1=float(B * H * W), eps=1e-5, momentum=0.1,
                              gamma1=g1.data_ptr(), beta1=b1.data_ptr(), mean1=mu.data_ptr(), rstd1=rs.data_ptr(), run_mean1=rm.data_ptr() if train else None,
                              run_var1=rv.data_ptr() if train else None, nbt1=nbt.data_ptr() if train else None, packed=packed.data_ptr(), bias2=bias2.data_ptr(),
                              y2=y2.data_ptr(), stat2=rows[1].data_ptr() if train else None)


res = {"B": B}
res["pack_us"] = ev(lambda: chk(L.eegclip_cstack_pack(Ws.data_ptr(), packed.data_ptr(), H, st)))
res["stats1_us"] = ev(lambda: chk(L.eegclip_cstack_stats1(x.data_ptr(), 64 * 250, 250, w25.data_ptr(), bias1.data_ptr(), rows[0].data_ptr(), B, H, st)))
for name, d in (("fwd_us", fwd_desc(False)), ("fwd_eval_us", fwd_desc(False, False))):
    res[name] = ev(lambda: chk(L.eegclip_cstack_fwd(ctypes.byref(d), st)))
# backward
packed_t = torch.empty(int(L.eegclip_cstack_packed_t_bytes(H)) // 2, dtype=torch.bfloat16, device=dev)
rows3 = torch.empty(B, 80, dtype=torch.float64, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
dx, dw25, dWs = torch.zeros(B, 64, 250, device=dev), torch.zeros(40, 25, device=dev), torch.zeros(C, C, H, device=dev)
dwp = torch.empty(int(L.eegclip_cstack_bwd_workspace_floats(B)), device=dev)
w2ws = torch.empty(int(L.eegclip_cstack_bwd_w2_workspace_floats(B, H)), device=dev)
bd = _abi.CstackBwdDesc(B=B, H=H, x=x.data_ptr(), xs_b=64 * 250, xs_h=250, w25=w25.data_ptr(), bias1=bias1.data_ptr(), mean1=mu.data_ptr(), rstd1=rs.data_ptr(),
                        gamma1=g1.data_ptr(), beta1=b1.data_ptr(), packed_t=packed_t.data_ptr(), dy2=dy2.data_ptr(), rows_out=rows3.data_ptr(), stat=rows3.data_ptr(),
                        nstat=B, count=float(B * H * W), stat_local=None, nstat_local=0, dgamma=dg.data_ptr(), dbeta=db.data_ptr(), dx=dx.data_ptr(),
                        dw_partials=dwp.data_ptr(), dw25=dw25.data_ptr())
res["pack_t_us"] = ev(lambda: chk(L.eegclip_cstack_pack_t(Ws.data_ptr(), packed_t.data_ptr(), H, st)))
res["bwd_stats_us"] = ev(lambda: chk(L.eegclip_cstack_bwd_stats(ctypes.byref(bd), st)))
res["bwd_apply_us"] = ev(lambda: chk(L.eegclip_cstack_bwd_apply(ctypes.byref(bd), st)))
res["bwd_w2_us"] = ev(lambda: chk(L.eegclip_cstack_bwd_w2(x.data_ptr(), 64 * 250, 250, w25.data_ptr(), bias1.data_ptr(), mu.data_ptr(), rs.data_ptr(), g1.data_ptr(),
                                                             b1.data_ptr(), dy2.data_ptr(), dWs.data_ptr(), w2ws.data_ptr(), B, H, st)))
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
