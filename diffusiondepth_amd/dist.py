"""Data-parallel plumbing for the hot path: the DDIM loop shards by independent images, so
inference needs NO data-path collective -- one process per GPU, each with its own HipDenoiser handle and
hipGraph.  What the reference does across ranks on this path (reference src/main.py:72-86,148,321-323):
  * DistributedSampler shards the image indices by rank              -> shard_indices()
  * rank 0 loads the checkpoint, apex DDP broadcasts parameters      -> broadcast_state_dict()
  * metrics are only logged for rank 0's shard (no reduction)        -> reduce_sums() (we do reduce)
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
