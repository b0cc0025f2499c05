"""Data-parallel glue: one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on ROCm).

The contrastive batch is partitioned row-wise across ranks; the only data-path exchange is the all-gather of the
embeddings (and targets) inside ClipLoss so the global negatives are preserved (+ the reduce-scatter of the gathered
embedding gradients), then ONE all-reduce of the flat parameter-gradient buffer (12.8 MB for ATMS).  With
ClipLoss(local_loss=True, gather_with_grad=True) every rank's gradient is W x its share of the single-process gradient
(SURVEY.md section 8e), so averaging over ranks reproduces the single-process step.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world_size).  A single process (WORLD_SIZE unset or 1) needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # EEGCLIP_DIST_BACKEND=gloo: several ranks on ONE GPU (functional check of the data-parallel step where RCCL would refuse duplicate devices)
        backend = backend or os.environ.get("EEGCLIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def average_flat_grads(flat_grad, engine=None):
    """all-reduce(mean) of one flat gradient buffer -- large RCCL collectives instead of ~40 small ones.  If the encoder's backward already
    started the all-reduce of its early bucket (engine.early_work: conv stack + head, 10.5 of 12.8 MB, hidden under the transformer's backward),
    wait for it and reduce only the two remaining pieces."""
    W = world_size()
    if W > 1:
        work = getattr(engine, "early_work", None) if engine is not None else None
        if work is not None:
            a0, a1 = engine.early_bucket
            engine.early_work = None
            if a0 > 0:
                dist.all_reduce(flat_grad[:a0], op=dist.ReduceOp.SUM)
            if a1 < flat_grad.numel():
                dist.all_reduce(flat_grad[a1:], op=dist.ReduceOp.SUM)
            work.wait()
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        flat_grad.mul_(1.0 / W)
    return flat_grad


def configure_loss_for_world(loss_mod, rank, world):
    """Switch a ClipLoss to the row-sharded, gradient-carrying gather mode used for data-parallel training."""
    loss_mod.rank, loss_mod.world_size = rank, world
    if world > 1:
        loss_mod.local_loss, loss_mod.gather_with_grad = True, True
    return loss_mod
