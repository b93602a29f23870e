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
    zeros so that the collective sequence is identical on every rank; a parameter NO rank has a gradient for keeps .grad = None.
    Returns the number of collectives issued."""
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
        # one extra element per parameter: 1 where this rank has a gradient.  After the SUM, 0 means NO rank produced one (a branch
        # that never executes, e.g. the HAHI attention parameters or convup_fp): its .grad stays None, as under the reference's apex
        # DDP, instead of becoming zeros that weight decay / Adam would then act on.
        flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in group], dtype=torch.float32, device=dev)
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in group] + [flags]).to(dev)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        n_coll += 1
        used = flat[-len(group):].tolist()
        if average:
            flat /= world
        off = 0
        for p, u in zip(group, used):
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is not None:
                p.grad.copy_(g)
            elif u > 0:
                p.grad = g.clone()
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


class OverlappedGradReducer:
    """Gradient averaging of data-parallel training OVERLAPPED with the backward pass (what apex DDP does for the reference,
    src/main.py:106-114,148; SURVEY.md 8e: the one exchange step per iteration).

    Parameters are cut into flat buckets of ~bucket_bytes in REVERSE registration order -- the order in which autograd finishes
    them: the depth head (whose gradients come out of dd_denoise_backward in one piece) is reduced while the backbone's backward is
    still running.  A bucket's all-reduce is issued asynchronously from the post-accumulate-grad hook of its last parameter, on the
    collective's own stream (RCCL) -- so it overlaps the remaining backward kernels -- and ``finish()`` waits, averages and writes the
    results back into ``.grad``.  Buckets are sized for xGMI (point-to-point links, per-link bandwidth bound): few and large.

        reducer = OverlappedGradReducer(model.parameters())
        loss.backward()            # collectives start as buckets fill
        reducer.finish()           # before optimizer.step()

    Parameters that received no gradient in this iteration (unused branches, e.g. the never-executed attention parameters of the
    HAHI neck) are reduced as zeros so that every rank issues the same collectives in the same order.  Buckets launch strictly in
    order, so a never-ready parameter would hold back every bucket behind it until finish(): the first finish() therefore learns
    (and agrees across ranks on) the set of parameters no rank produced a gradient for, and later iterations treat them as ready
    from the start; if one of them does receive a gradient later, finish() raises instead of silently dropping it
    (``relearn_unused()`` starts over); their .grad is left untouched (None stays None, as under apex DDP).

    Gradient accumulation over several backward passes: wrap all but the last micro-batch in ``no_sync()`` (hooks idle, gradients
    accumulate locally, nothing is sent) -- the last backward then reduces the accumulated sums, overlapped as usual.  Without
    ``no_sync()`` the result is still correct: a bucket whose parameters receive another gradient after its all-reduce was issued is
    marked stale and reduced AGAIN in finish() from the accumulated ``.grad`` (the first transfer is wasted, not the gradient).
    Not distributed (no process group, or one of a single rank without ``force_single_rank``): every method is a no-op."""

    force_single_rank = False      # class-wide switch (tests), like SyncBatchNorm.force_sync

    def __init__(self, params, bucket_bytes: int = 64 << 20, average: bool = True, force_single_rank: bool = False):
        """force_single_rank: take the collective path with a process group of ONE rank too (hooks, async all-reduce on the collective's own
        stream, finish()): a one-GPU box then executes exactly the code N ranks execute (bench.py --mode train-dp under a launcher,
        tests/test_zz_gpu_dist.py); the sums over one rank are the identity."""
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_single_rank or self.force_single_rank)
        self.buckets: List[List[int]] = []
        cur, nbytes = [], 0
        for i in reversed(range(len(self.params))):
            n = self.params[i].numel() * 4
            if cur and nbytes + n > bucket_bytes:
                self.buckets.append(cur)
                cur, nbytes = [], 0
            cur.append(i)
            nbytes += n
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {i: b for b, idx in enumerate(self.buckets) for i in idx}
        self._pending = [len(idx) for idx in self.buckets]
        self._ready = [False] * len(self.params)
        self._work = [None] * len(self.buckets)
        self._flat = [None] * len(self.buckets)
        self._unused = None                # indices no rank had a gradient for in the first iteration (None = not learned yet)
        self._violation = None
        self._next = 0                     # buckets are launched strictly in order (identical collective sequence on all ranks)
        self._stale = [False] * len(self.buckets)   # a gradient arrived after the bucket's all-reduce was issued (second backward pass)
        self._sync = True                  # False inside no_sync()
        self.launched_in_backward = 0      # statistics: buckets whose all-reduce started from a hook (i.e. overlapped)
        self._handles = []
        if self.active:
            for i, p in enumerate(self.params):
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(_param):
            if self._unused is not None and i in self._unused:
                self._violation = i        # declared unused after the first iteration, yet it has a gradient now: finish() raises
                return
            if not self._sync:
                return                     # no_sync(): accumulate locally, send nothing
            b = self._bucket_of[i]
            if self._ready[i]:
                # another backward pass without no_sync(): .grad now holds more than what was (or will be) flattened
                if b < self._next:
                    self._stale[b] = True  # already on the wire: finish() reduces this bucket again from the accumulated .grad
                return
            self._ready[i] = True
            self._pending[b] -= 1
            self._launch_ready(from_hook=True)
        return hook

    def _launch(self, b):
        group = [self.params[i] for i in self.buckets[b]]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in group])
        self._flat[b] = flat
        self._work[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _launch_ready(self, from_hook):
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self.launched_in_backward += int(from_hook)
            self._next += 1

    def finish(self) -> int:
        """Waits for every bucket (launching, with zeros for missing gradients, those that never filled), writes the averaged
        gradients back and re-arms the reducer for the next iteration.  Returns the number of collectives of this iteration."""
        if not self.active:
            return 0
        if self._violation is not None:
            i, self._violation = self._violation, None
            raise RuntimeError(f"OverlappedGradReducer: parameter #{i} (shape {tuple(self.params[i].shape)}) had no gradient on any rank in "
                               "the first iteration but received one now; its bucket was already reduced without it -- call "
                               "relearn_unused() when the set of trained parameters changes")
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        world = dist.get_world_size()
        if self._unused is None:           # agree on the parameters nobody produced a gradient for (MAX over ranks of "was ready")
            used = torch.tensor([1.0 if r else 0.0 for r in self._ready])
            dev = self.params[0].device if self.params else torch.device("cpu")
            used = used.to(dev)
            dist.all_reduce(used, op=dist.ReduceOp.MAX)
            self._unused = {i for i, u in enumerate(used.tolist()) if u == 0.0}
        # stale buckets (same on every rank only if every rank ran the same number of backward passes -- agree on the union)
        st = torch.tensor([1.0 if x else 0.0 for x in self._stale], device=self.params[0].device if self.params else torch.device("cpu"))
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
        n = len(self.buckets)
        for b, x in enumerate(st.tolist()):
            if x > 0:
                self._work[b].wait()       # the superseded transfer
                self._launch(b)            # .grad has not been written back yet: it still holds the locally accumulated sum
                n += 1
        for b, idx in enumerate(self.buckets):
            self._work[b].wait()
            flat = self._flat[b]
            if self.average:
                flat /= world
            off = 0
            for i in idx:
                p = self.params[i]
                if i not in self._unused:          # nobody trained it: leave .grad alone (None stays None)
                    g = flat[off:off + p.numel()].view_as(p).to(p.dtype)
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.copy_(g)
                off += p.numel()
        self._rearm()
        return n

    def no_sync(self):
        """Context for the micro-batches of a gradient-accumulation step that are NOT the last one (the name DDP uses)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._sync = self._sync, False
            try:
                yield self
            finally:
                self._sync = prev
        return ctx()

    def _rearm(self):
        n = len(self.buckets)
        self._pending = [len(idx) for idx in self.buckets]
        self._ready = [False] * len(self.params)
        for i in (self._unused or ()):     # known-unused parameters never hold a bucket back
            self._ready[i] = True
            self._pending[self._bucket_of[i]] -= 1
        self._work = [None] * n
        self._flat = [None] * n
        self._stale = [False] * n
        self._next = 0

    def relearn_unused(self):
        """Forget which parameters are unused (call on every rank when the trained parameter set changes, e.g. after unfreezing)."""
        self._unused, self._violation = None, None
        self._rearm()

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


# ------------------------------------------------------------------------------------------------------------------------------
# Synchronised BatchNorm for data-parallel training.  The reference converts EVERY BatchNorm of the network before wrapping it in
# apex DDP (`net = apex.parallel.convert_syncbn_model(net)`, reference src/main.py:128): in .train() the FPN, the latent codec and the
# backbone normalise with the statistics of the GLOBAL batch (all ranks), forward and backward.  These modules stay PyTorch modules in
# .train() mode here, so the exchange is two small all-reduces per layer on whatever backend the process group
# has (RCCL on the GPUs; gloo in the CPU tests, which torch.nn.SyncBatchNorm refuses).
# ------------------------------------------------------------------------------------------------------------------------------
class _SyncBatchNormFn(torch.autograd.Function):
    """y = (x - mean_G) * invstd_G * w + b with mean / biased variance over the batch and spatial dims of ALL ranks.
    Forward exchange: [n * mean, n * (var + mean^2), n] per channel (summed); backward exchange: [sum dy, sum dy * (x - mean_G)]."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        n = x.numel() // C
        var, mean = torch.var_mean(x.float(), dim=dims, unbiased=False)
        packed = torch.cat([mean * n, (var + mean * mean) * n, torch.full((1,), float(n), device=x.device)])
        dist.all_reduce(packed, group=group)
        N = packed[-1]
        mean_g = packed[:C] / N
        var_g = (packed[C:2 * C] / N - mean_g * mean_g).clamp_min_(0.0)
        invstd = torch.rsqrt(var_g + eps)
        if running_mean is not None:
            with torch.no_grad():      # as nn.BatchNorm2d: running_var takes the unbiased estimate
                running_mean.mul_(1 - momentum).add_(mean_g.to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1 - momentum).add_((var_g * (N / (N - 1).clamp_min(1.0))).to(running_var.dtype), alpha=momentum)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = (x.float() - mean_g.view(shape)) * invstd.view(shape)
        w = weight.float().view(shape) if weight is not None else None
        y = xhat * w if w is not None else xhat
        if bias is not None:
            y = y + bias.float().view(shape)
        ctx.save_for_backward(xhat, invstd, weight)
        ctx.group, ctx.dims, ctx.shape, ctx.xdtype = group, dims, shape, x.dtype
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        xhat, invstd, weight = ctx.saved_tensors
        dims, shape = ctx.dims, ctx.shape
        gy = gy.float()
        sum_dy = gy.sum(dims)
        sum_dy_xhat = (gy * xhat).sum(dims)
        C = sum_dy.numel()
        n = gy.numel() // C
        packed = torch.cat([sum_dy, sum_dy_xhat, torch.full((1,), float(n), device=gy.device)])
        dist.all_reduce(packed, group=ctx.group)
        N = packed[-1]
        w = weight.float() if weight is not None else torch.ones_like(invstd)
        # dx = w * invstd * (dy - mean_G(dy) - xhat * mean_G(dy * xhat))
        gx = (gy - (packed[:C] / N).view(shape) - xhat * (packed[C:2 * C] / N).view(shape)) * (w * invstd).view(shape)
        gw = sum_dy_xhat.to(weight.dtype) if weight is not None and ctx.needs_input_grad[1] else None      # LOCAL sums: the gradient exchange
        gb = sum_dy.to(weight.dtype) if ctx.needs_input_grad[2] else None                                     # averages parameters' gradients
        return gx.to(ctx.xdtype), gw, gb, None, None, None, None, None


class SyncBatchNorm(torch.nn.modules.batchnorm._BatchNorm):
    """BatchNorm over the global batch in .train() under an initialised process group of more than one rank; plain batch_norm otherwise
    (eval mode, a single process).  Same parameters / buffers / state-dict keys as nn.BatchNorm{1,2,3}d."""

    force_sync = False      # tests: take the exchanging path with a single rank too (exercises the collectives on a one-GPU box)

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.process_group = process_group

    def _check_input_dim(self, input):
        if input.dim() < 2:
            raise ValueError(f"expected at least 2D input (got {input.dim()}D input)")

    def forward(self, x):
        self._check_input_dim(x)
        sync = self.training and dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.process_group) > 1 or self.force_sync)
        if not sync:
            return super().forward(x)
        momentum = 0.0 if self.momentum is None else self.momentum
        if self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:
                momentum = 1.0 / float(self.num_batches_tracked)
        return _SyncBatchNormFn.apply(x, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                      self.running_var if self.track_running_stats else None, self.eps, momentum, self.process_group)


def convert_sync_batchnorm(module: torch.nn.Module, process_group=None) -> torch.nn.Module:
    """Every nn.BatchNorm*d of `module` becomes a SyncBatchNorm holding the SAME parameter and buffer tensors (optimizers built before the
    conversion stay valid; state-dict keys unchanged).  Replaces `apex.parallel.convert_syncbn_model(net)` (reference src/main.py:128)."""
    out = module
    if isinstance(module, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(module, SyncBatchNorm):
        out = SyncBatchNorm(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats, process_group)
        if module.affine:
            out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var, out.num_batches_tracked = module.running_mean, module.running_var, module.num_batches_tracked
        out.training = module.training
    for name, child in module.named_children():
        new = convert_sync_batchnorm(child, process_group)
        if new is not child:
            setattr(out, name, new)      # (a converted parent has no children: _BatchNorm modules are leaves)
    return out
