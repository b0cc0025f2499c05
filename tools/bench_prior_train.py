"""configs[3] alone: the diffusion-prior training step at batch 1024 (bench.py's secondary line) -- run under rocprofv3 --kernel-trace for the launch list;
also the HOST time of a step (enqueue without waiting for the GPU)"""
import contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
out = bench._sec_prior_train()
from eeg_image_decode_amd.prior import DiffusionPriorUNet, Pipe
g = torch.Generator().manual_seed(0)
c, h = torch.randn(1024, 1024, generator=g).cuda(), torch.randn(1024, 1024, generator=g).cuda()
pipe = Pipe(DiffusionPriorUNet(cond_dim=1024, dropout=0.1), device="cuda")
dl = [{"c_embedding": c, "h_embedding": h}] * 6
with contextlib.redirect_stdout(io.StringIO()):
    pipe.train(dl, num_epochs=1, learning_rate=1e-3)
    pipe.cond_drop_prob = 1.0
    pipe.train(dl[:2], num_epochs=1, learning_rate=1e-3)
    pipe.cond_drop_prob = 0.1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # one long epoch: a single host sync (the loss readout) at its end
    pipe.train(dl * 8, num_epochs=1, learning_rate=1e-3)
    t1 = time.perf_counter()
torch.cuda.synchronize()
out["wall_ms_per_step_one_epoch_of_48"] = round(1e3 * (t1 - t0) / 48, 3)
print(json.dumps(out))
