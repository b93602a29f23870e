"""GPU (`-m gpu`): the data-parallel plumbing on the RCCL backend with ONE rank (all a one-GPU box offers): process group on `nccl`, the
bucketed gradient all-reduce and the SyncBatchNorm exchange on device tensors.  Runs in a subprocess (the process group is global state)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.environ["DD_ROOT"])
import torch.distributed as dist
from diffusiondepth_amd import dist as ddist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
torch.manual_seed(5)
def net():
    torch.manual_seed(5)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1)).cuda().train()
ref, syn = net(), ddist.convert_sync_batchnorm(net())
ddist.SyncBatchNorm.force_sync = True                       # one rank, but through the all-reduces
x = torch.randn(5, 3, 24, 40, device="cuda") * 2 + 1
xr, xs = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
(ref(xr) ** 2).mean().backward()
(syn(xs) ** 2).mean().backward()
ddist.allreduce_gradients(syn.parameters())                 # (a single rank has nothing to exchange: returns without a collective)
flat = torch.cat([p.grad.reshape(-1) for p in syn.parameters()]); want = flat.clone()
dist.all_reduce(flat); assert torch.equal(flat, want)       # ... so one explicit RCCL all-reduce of the flat gradients: identity over 1 rank
red = ddist.OverlappedGradReducer(list(syn.parameters()))
close = lambda a, b: float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
assert close(xs.grad, xr.grad)
assert all(close(a.grad, b.grad) for a, b in zip(syn.parameters(), ref.parameters()))
assert all(close(a.float(), b.float()) for a, b in zip(syn.buffers(), ref.buffers()))
for p in syn.parameters(): p.grad = None
(syn(xs.detach()) ** 2).mean().backward(); red.finish(); red.close()
assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in syn.parameters())
dist.barrier(); dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


def test_rccl_process_group_gradient_allreduce_and_sync_batchnorm_on_one_rank():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", DD_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
