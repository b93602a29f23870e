"""CPU, world_size 2, gloo: the N>1 host path (image sharding, parameter broadcast, exact metric reduction).
The hot path itself shards by independent images and has no data-path collective."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusiondepth_amd import dist as ddist
from diffusiondepth_amd import synth


def test_shards_partition_the_items():
    for n in (0, 1, 7, 8, 9, 33):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in ddist.shard_indices(n, r, world))
            assert seen == list(range(n))
            per = {len(ddist.shard_indices(n, r, world, pad=True)) for r in range(world)}
            assert len(per) == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = ddist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # rank 0 "loads the checkpoint"; everybody else starts from garbage and receives the broadcast
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(7240).items()}
    if rank != 0:
        for v in sd.values():
            v.fill_(-1.0)
    ddist.broadcast_state_dict(sd)
    ref = synth.make_state_dict(7240)
    assert all(np.array_equal(sd[k].numpy(), ref[k]) for k in ref)
    # each rank evaluates its shard of 5 synthetic images; reduced metric sums must equal the single-process ones
    n_img = 5
    sums = torch.zeros(4, dtype=torch.float64)
    for i in ddist.shard_indices(n_img, rank, world):
        gt = torch.from_numpy(synth.make_gt_depth(100 + i, 1, 16, 24))
        pred = gt + torch.from_numpy(np.random.RandomState(200 + i).standard_normal(gt.shape).astype(np.float32)) * 0.1
        sums += ddist.depth_metric_sums(pred, gt)
    tot = ddist.reduce_sums(sums)
    if rank == 0:
        torch.save(tot, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_metric_reduction(tmp_path):
    out = str(tmp_path / "tot.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    tot = torch.load(out)
    ref = torch.zeros(4, dtype=torch.float64)
    for i in range(5):
        gt = torch.from_numpy(synth.make_gt_depth(100 + i, 1, 16, 24))
        pred = gt + torch.from_numpy(np.random.RandomState(200 + i).standard_normal(gt.shape).astype(np.float32)) * 0.1
        ref += ddist.depth_metric_sums(pred, gt)
    assert torch.allclose(tot, ref, rtol=1e-12, atol=0)
    m = ddist.finalize_metrics(tot)
    assert 0.05 < m["rmse"] < 0.2 and m["n"] > 0


def _grad_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ddist.init_from_env("gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in [(64, 16, 3, 3), (64,), (256, 64, 3, 3), (1280, 256), (16,), (7,)]]
    for k, p in enumerate(params):
        if (k == 4 and rank == 1) or k == 5:
            continue                                   # 4: no gradient on this rank;  5: no gradient on ANY rank (a never-executed branch)
        p.grad = torch.full_like(p, float(rank + 1) * (k + 1))
    n = ddist.allreduce_gradients(params, bucket_bytes=1 << 20)      # 1 MiB buckets -> several collectives
    assert params[5].grad is None                      # stays None on every rank, as under the reference's apex DDP (no zeros for Adam / decay)
    if rank == 0:
        torch.save({"n": n, "g": [p.grad.clone() for p in params[:5]]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce(tmp_path):
    """Bucketed gradient averaging (the one exchange step of data-parallel training): rank r holds grad = (r+1)(k+1) for
    parameter k -> the average is 1.5 (k+1); the last parameter has a gradient on rank 0 only -> 2.5."""
    out = str(tmp_path / "g.pt")
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["n"] >= 2
    for k, g in enumerate(r["g"]):
        want = 1.5 * (k + 1) if k < 4 else 2.5
        assert torch.allclose(g, torch.full_like(g, want)), k


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ddist.init_from_env("gloo")
    torch.manual_seed(0)                                     # same initial weights on both ranks
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(8, 1, 3, padding=1))
    unused = torch.nn.Parameter(torch.ones(5))               # a parameter no loss term touches (e.g. the HAHI attention parameters)
    params = list(net.parameters()) + [unused]
    red = ddist.OverlappedGradReducer(params, bucket_bytes=1024)      # tiny buckets -> several collectives, launched during backward
    res = {}
    for it in range(2):                                       # two iterations: the reducer re-arms itself
        for p in params:
            p.grad = None
        x = torch.from_numpy(np.random.RandomState(10 * it + rank).standard_normal((2, 3, 12, 12)).astype(np.float32))
        net(x).square().mean().backward()
        res[f"launched_in_backward_{it}"] = red.launched_in_backward
        res[f"n_{it}"] = red.finish()
        assert unused.grad is None                             # nobody trained it: not materialised as zeros
        res[f"g_{it}"] = [p.grad.clone() for p in params[:-1]]
    # gradient accumulation, two micro-batches per step: (a) the first under no_sync(), (b) both plain -- the buckets that
    # were already on the wire when the second backward ran are reduced again in finish().  Both must give the rank-average of the SUMS.
    for mode in ("no_sync", "plain"):
        for p in params:
            p.grad = None
        for mb in range(2):
            x = torch.from_numpy(np.random.RandomState(100 + 10 * mb + rank).standard_normal((2, 3, 12, 12)).astype(np.float32))
            if mode == "no_sync" and mb == 0:
                with red.no_sync():
                    net(x).square().mean().backward()
            else:
                net(x).square().mean().backward()
        res[f"n_acc_{mode}"] = red.finish()
        res[f"g_acc_{mode}"] = [p.grad.clone() for p in params[:-1]]
    red.close()
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_overlapped_gradient_reducer(tmp_path):
    """OverlappedGradReducer: buckets are all-reduced from autograd hooks while backward is still running; the result equals the
    average of the two ranks' gradients computed in one process; a parameter nobody trained keeps .grad None; the reducer re-arms;
    gradient accumulation (with and without no_sync()) reduces the accumulated sums."""
    out = str(tmp_path / "o.pt")
    mp.spawn(_overlap_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(8, 1, 3, padding=1))
    for it in range(2):
        want = [torch.zeros_like(p) for p in net.parameters()]
        for rank in range(2):
            net.zero_grad()
            x = torch.from_numpy(np.random.RandomState(10 * it + rank).standard_normal((2, 3, 12, 12)).astype(np.float32))
            net(x).square().mean().backward()
            for w, p in zip(want, net.parameters()):
                w += p.grad / 2
        got = r[f"g_{it}"]
        assert r[f"n_{it}"] >= 3
        for k, (g, w) in enumerate(zip(got, want)):
            assert torch.allclose(g, w, rtol=1e-5, atol=1e-7), (it, k)
    # iteration 0: the unused parameter sits in the first bucket and holds the in-order launches back until finish(); finish() learns
    # (across ranks) that nobody has a gradient for it, so in iteration 1 EVERY bucket starts its all-reduce from a hook inside backward
    assert r["launched_in_backward_0"] == 0
    assert r["launched_in_backward_1"] - r["launched_in_backward_0"] == r["n_1"]
    # accumulation over two micro-batches: average over ranks of (g_mb0 + g_mb1)
    want = [torch.zeros_like(p) for p in net.parameters()]
    for rank in range(2):
        net.zero_grad()
        for mb in range(2):
            x = torch.from_numpy(np.random.RandomState(100 + 10 * mb + rank).standard_normal((2, 3, 12, 12)).astype(np.float32))
            net(x).square().mean().backward()
        for w, p in zip(want, net.parameters()):
            w += p.grad / 2
    for mode in ("no_sync", "plain"):
        for k, (g, w) in enumerate(zip(r[f"g_acc_{mode}"], want)):
            assert torch.allclose(g, w, rtol=1e-5, atol=1e-7), (mode, k)
    assert r["n_acc_plain"] > r["n_acc_no_sync"]            # the plain form paid for re-sending the stale buckets


def test_overlapped_reducer_is_a_noop_without_a_process_group():
    p = torch.nn.Parameter(torch.ones(3))
    red = ddist.OverlappedGradReducer([p])
    p.sum().backward()
    assert red.finish() == 0 and torch.equal(p.grad, torch.ones(3))


# ---- SyncBatchNorm (reference src/main.py:128: apex.parallel.convert_syncbn_model before DDP) ----------------------------------------
def _bn_net():
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 6, 3, padding=1), torch.nn.BatchNorm2d(6), torch.nn.ReLU(),
                              torch.nn.ConvTranspose2d(6, 5, 2, stride=2), torch.nn.BatchNorm2d(5), torch.nn.ReLU(), torch.nn.Conv2d(5, 2, 1))
    with torch.no_grad():
        for m in net:
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    return net


def _bn_batch():
    g = torch.Generator().manual_seed(11)
    return torch.randn(6, 3, 9, 7, generator=g) * 2.0 + 0.5, torch.randn(6, 2, 18, 14, generator=g)


def _syncbn_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ddist.init_from_env("gloo")
    net = _bn_net()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)                   # built BEFORE the conversion: must keep pointing at the live tensors
    net = ddist.convert_sync_batchnorm(net).train()
    assert sum(isinstance(m, ddist.SyncBatchNorm) for m in net.modules()) == 2 and not any(type(m) is torch.nn.BatchNorm2d for m in net.modules())
    x, tgt = _bn_batch()
    sl = slice(0, 4) if rank == 0 else slice(4, 6)                   # UNEVEN shards: the statistics weigh ranks by their element counts
    xr = x[sl].clone().requires_grad_(True)
    # each rank's loss is a SUM over its samples divided by the global batch: the average over ranks x world == the full-batch mean loss
    loss = ((net(xr) - tgt[sl]) ** 2).sum() / x.shape[0]
    loss.backward()
    ddist.allreduce_gradients(net.parameters(), average=False)
    opt.step()
    eval_out = net.eval()(x[:2])                                       # running statistics in use; no collective in eval mode
    torch.save({"gx": xr.grad, "grads": [p.grad.clone() for p in net.parameters()], "params": [p.detach().clone() for p in net.parameters()],
                "buffers": [b.clone() for b in net.buffers()], "eval": eval_out.detach()}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sync_batchnorm_equals_one_process_on_the_whole_batch(tmp_path):
    """convert_sync_batchnorm: two ranks with uneven shards of a batch == one process with plain BatchNorm on the whole batch -- outputs
    of the normalised net, input gradients, parameter gradients after the gradient exchange, the SGD update, running statistics."""
    out = str(tmp_path / "sbn.pt")
    mp.spawn(_syncbn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    net = _bn_net().train()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x, tgt = _bn_batch()
    xr = x.clone().requires_grad_(True)
    (((net(xr) - tgt) ** 2).sum() / x.shape[0]).backward()
    want_grads = [p.grad.clone() for p in net.parameters()]
    opt.step()
    want_eval = net.eval()(x[:2]).detach()
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    # (the conv biases in front of a BatchNorm have gradient 0: what is left there is the cancellation noise of sums of magnitude ~100)
    close = lambda a, b: float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    assert close(torch.cat([r0["gx"], r1["gx"]]), xr.grad)
    for r in (r0, r1):
        assert all(close(a, b) for a, b in zip(r["grads"], want_grads))
        assert all(close(a, b.detach()) for a, b in zip(r["params"], net.parameters()))
        assert all(close(a.float(), b.float()) for a, b in zip(r["buffers"], net.buffers()))
        assert close(r["eval"], want_eval)


def test_sync_batchnorm_is_plain_batchnorm_in_one_process():
    net, ref = ddist.convert_sync_batchnorm(_bn_net()).train(), _bn_net().train()
    x, _ = _bn_batch()
    assert torch.allclose(net(x), ref(x), atol=1e-6) and list(net.state_dict()) == list(ref.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(net.buffers(), ref.buffers()))
