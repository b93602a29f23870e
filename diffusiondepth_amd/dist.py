"""Data-parallel plumbing for the hot path: the DDIM loop shards by independent images, so
inference needs NO data-path collective -- one process per GPU, each with its own HipDenoiser handle and
hipGraph.  What the reference does across ranks on this path (reference src/main.py:72-86,148,321-323):
  * DistributedSampler shards the image indices by rank              -> shard_indices()
  * rank 0 loads the checkpoint, apex DDP broadcasts parameters      -> broadcast_state_dict()
  * metrics are only logged for rank 0's shard (no reduction)        -> reduce_sums() (we do reduce)
  * training: apex DDP averages the gradients of all parameters      -> allreduce_gradients(): the ONE exchange step per
    iteration, as a few large flat buckets -- xGMI is point-to-point (7 links x ~153 GB/s per GPU), ring
    collectives are per-link bound, so bucket size is chosen for bandwidth (default 64 MiB), not for NVSwitch latency
Backend 'nccl' == RCCL on ROCm (GPU tensors); 'gloo' on CPU is used by the tests.
"""
from __future__ import annotations

import os
from typing import Dict, List

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as set by torch.distributed.run.  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shard_indices(n_items: int, rank: int, world: int, pad: bool = False) -> List[int]:
    """Rank r takes items r, r+world, ... (DistributedSampler order without shuffling).  With pad=True the
    shards are padded to equal length by wrapping around (what DistributedSampler does); otherwise the union
    of all shards is exactly range(n_items)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    idx = list(range(rank, n_items, world))
    if pad and n_items > 0:
        per = (n_items + world - 1) // world
        k = 0
        while len(idx) < per:
            idx.append((rank + k * world) % n_items)
            k += 1
    return idx


def reduce_sums(sums: torch.Tensor) -> torch.Tensor:
    """All-reduce (SUM) of a small tensor of metric numerators/denominators; identity when not distributed."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        sums = sums.clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums


def allreduce_gradients(params, bucket_bytes: int = 64 << 20, average: bool = True) -> int:
    """Averages .grad of `params` over all ranks (gradient all-reduce of data-parallel training, reference
    src/main.py:106-114 via apex DDP).  Gradients are packed into flat buckets of ~bucket_bytes in parameter order (the
    same order on every rank), one all_reduce per bucket, and unpacked in place.  Parameters without .grad contribute
    zeros so that the collective sequence is identical on every rank.  Returns the number of collectives issued."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return 0
    world = dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    n_coll = 0
    i = 0
    while i < len(params):
        j, nbytes = i, 0
        while j < len(params) and (j == i or nbytes + params[j].numel() * 4 <= bucket_bytes):
            nbytes += params[j].numel() * 4
            j += 1
        group = params[i:j]
        dev = group[0].device
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in group]).to(dev)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        n_coll += 1
        if average:
            flat /= world
        off = 0
        for p in group:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        i = j
    return n_coll


def broadcast_state_dict(sd: Dict[str, torch.Tensor], src: int = 0) -> Dict[str, torch.Tensor]:
    """Parameter broadcast from `src` (the reference relies on apex DDP doing this at wrap time,
    src/main.py:106-114,148).  Tensors are updated in place, in sorted-key order on every rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for k in sorted(sd):
            dist.broadcast(sd[k], src=src)
    return sd


def depth_metric_sums(pred: torch.Tensor, gt: torch.Tensor, t_valid: float = 1e-4) -> torch.Tensor:
    """Per-shard sums behind RMSE / MAE / REL over valid pixels (reference src/metric/diffusion_dcbase_metric.py:31-93):
    returns [sum sq err, sum abs err, sum rel err, n_valid] so that shards can be reduced exactly."""
    mask = gt > t_valid
    d = (pred - gt)[mask]
    g = gt[mask]
    return torch.stack([(d * d).sum(), d.abs().sum(), (d.abs() / g).sum(), mask.sum().to(pred.dtype)]).double()


def finalize_metrics(sums: torch.Tensor) -> Dict[str, float]:
    n = max(float(sums[3]), 1.0)
    return {"rmse": float((sums[0] / n).sqrt()), "mae": float(sums[1] / n), "rel": float(sums[2] / n), "n": n}
