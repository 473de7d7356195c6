"""Multi-GPU: one process per GPU, batch sharded by rank, ONE collective per forward.

The forward has no cross-image operation, so ranks only exchange the packed
detections [B_local, Q, C+4] fp32 written contiguously by the head kernels.  On PyTorch-ROCm the
"nccl" backend is RCCL; the message is 304 KB .. 3.6 MB per rank (latency-bound over xGMI), hence
one all_gather_into_tensor of one contiguous buffer instead of per-tensor gathers.  The reference
itself never gathers detections (per-rank JSON files, train.py:820); this is the north-star's addition.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(global_batch, rank, world):
    """Contiguous split: rank r gets images [lo, hi).  Remainder images go to the first ranks."""
    q, r = divmod(global_batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_detections(det_local, group=None):
    """[B_local, Q, C+4] on every rank -> [sum B_local, Q, C+4] on every rank, rank order.
    Equal B_local on all ranks: one all_gather_into_tensor.  Ragged: padded to the max and trimmed."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return det_local
    world = dist.get_world_size(group)
    det_local = det_local.contiguous()
    sizes = [None] * world
    n = torch.tensor([det_local.shape[0]], device=det_local.device, dtype=torch.int64)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    sizes = [int(t.item()) for t in ns]
    if len(set(sizes)) == 1:
        out = torch.empty((world * sizes[0],) + tuple(det_local.shape[1:]), dtype=det_local.dtype, device=det_local.device)
        dist.all_gather_into_tensor(out, det_local, group=group)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(det_local.shape[1:]), dtype=det_local.dtype, device=det_local.device)
    pad[: det_local.shape[0]] = det_local
    out = torch.empty((world * mx,) + tuple(det_local.shape[1:]), dtype=det_local.dtype, device=det_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def gather_detections_equal(det_local, out=None, group=None):
    """Fast path used by bench.py: equal shards (known statically), no size exchange, optional
    preallocated output -> exactly one RCCL all-gather per forward."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return det_local
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * det_local.shape[0],) + tuple(det_local.shape[1:]), dtype=det_local.dtype,
                          device=det_local.device)
    dist.all_gather_into_tensor(out, det_local.contiguous(), group=group)
    return out
