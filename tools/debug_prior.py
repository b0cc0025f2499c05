import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn.functional as F
from eeg_image_decode_amd import synthetic as syn
from eeg_image_decode_amd.prior import DiffusionPriorUNet
from oracle import prior as oprior, loops as oloops
SEED = 20260926
m = DiffusionPriorUNet(cond_dim=1024, dropout=0.0)
state = syn.make_state(SEED + 20, oprior.prior_state_spec())
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
m = m.cuda().eval(); P = oloops.torch_state(state)
x = torch.from_numpy(syn.unit_features(SEED + 21, 6, tag="px") * 8.0); c = torch.from_numpy(syn.unit_features(SEED + 21, 6, tag="pc") * 32.0)
tt = torch.tensor([0, 5, 333, 999, 20, 980])
with torch.no_grad():
    out = m(x.cuda(), tt.cuda(), c.cuda())
ref = oprior.prior_unet_forward(P, x, tt, c)
print("out err", float((out.cpu() - ref).abs().max()))
b = m._engine().bufs[6]
temb = oprior.timestep_embedding(tt)
print("temb err", float((b["temb"].cpu() - temb).abs().max()))
lin0 = F.linear(x, P["input_layer.0.weight"], P["input_layer.0.bias"]); print("lin0", float((b["linI"].cpu() - lin0).abs().max()))
ln0 = F.layer_norm(lin0, (1024,), P["input_layer.1.weight"], P["input_layer.1.bias"]); print("ln0", float((b["lnI"].cpu() - ln0).abs().max()))
cur = F.silu(ln0); print("act0", float((b["actI"].cpu() - cur).abs().max()))
skips = []
for s in range(8):
    st = m._engine().stages[s]
    if st["dec"] is None: skips.append(cur)
    t1pre = F.linear(temb, P[st["t"] + "linear_1.weight"], P[st["t"] + "linear_1.bias"])
    print(s, "t1pre", float((b[f"t1pre{s}"].cpu() - t1pre).abs().max()), "t1act", float((b[f"t1act{s}"].cpu() - F.silu(t1pre)).abs().max()))
    xin = cur + F.linear(F.silu(t1pre), P[st["t"] + "linear_2.weight"], P[st["t"] + "linear_2.bias"]) + F.linear(c, P[st["c"] + "weight"], P[st["c"] + "bias"])
    print(s, "xin", float((b[f"xin{s}"].cpu() - xin).abs().max()))
    lin = F.linear(xin, P[st["l"] + "0.weight"], P[st["l"] + "0.bias"]); print(s, "lin", float((b[f"lin{s}"].cpu() - lin).abs().max()))
    ln = F.layer_norm(lin, (lin.shape[-1],), P[st["l"] + "1.weight"], P[st["l"] + "1.bias"]); print(s, "ln", float((b[f"ln{s}"].cpu() - ln).abs().max()))
    cur = F.silu(ln)
    if st["dec"] is not None: cur = cur + skips[-1 - st["dec"]]
    print(s, "act", float((b[f"act{s}"].cpu() - cur).abs().max()))
