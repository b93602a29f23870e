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


LANES_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.environ["DD_ROOT"])
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)        # the eager communicator FIRST: a data-parallel job's order
dist.barrier(); torch.cuda.synchronize()
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
H, W, B, T = 352, 1216, 4, 20
h, w = synth.latent_hw(H, W)
inp = synth.make_inputs(7240, B, h, w)
x_T, cond = torch.from_numpy(inp["x_T"]).to(dev), torch.from_numpy(inp["cond"]).to(dev)
def rate(probe, lanes):
    be = dda.HipDenoiser(dev); be.load_state_dict(synth.make_state_dict(7240)); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    be.set_option("lane_probe", probe); be.set_option("streams", lanes)
    x0 = torch.empty_like(x_T)
    for _ in range(3):
        be.denoise(x_T, cond, T, "f16r", out=x0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        be.denoise(x_T, cond, T, "f16r", out=x0)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10, be.counter("lane_overlap"), be.counter("lane_probe_retries"), x0.clone()
one, ov1, _, ref = rate(1, 1)
two, ov2, retries, got = rate(1, 2)
raw, ov0, _, _ = rate(0, 2)
print(f"LANES one {one:.3f} ms, two probed {two:.3f} ms (overlap {ov2}, retries {retries}), two unprobed {raw:.3f} ms (overlap {ov0})")
assert ov1 == -1 and ov0 == -1 and ov2 == 1, (ov1, ov0, ov2)
assert float((ref - got).abs().max()) <= 2e-2 * float(ref.abs().max())      # (bit-identical when both take the same conv3 tile form: tests/test_gpu_parity.py)
assert two < 0.99 * one, (one, two)                # concurrent lanes beat one lane ...
assert two <= raw * 1.02, (two, raw)               # ... and are never slower than the unprobed pair (which shares a hardware queue in this order: ~1.25x)
dist.destroy_process_group()
print("LANES-OK")
"""


def test_lanes_run_concurrently_in_a_process_whose_rccl_communicator_came_first():
    """Round 6: in a process that initialised an RCCL communicator eagerly BEFORE the library handle existed -- every rank of a data-parallel job -- the HIP
    runtime put a lane's new stream on the caller's own hardware queue (GPU_MAX_HW_QUEUES = 4 queues, the least-used one handed out once they exist) and the
    two lanes serialised: the KITTI step ran 23 % slower than concurrent lanes (profiles/r06_experiments.md section 9).  The library now probes a lane's
    stream for concurrency with the caller's when it creates it (dd_api.cpp: acquire_lane_stream) and replaces one that failed."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", DD_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", LANES_SCRIPT], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "LANES-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    import gpu_util
    gpu_util.record("lanes_after_rccl", line=[l for l in r.stdout.splitlines() if l.startswith("LANES one")][-1])


# ---- bench.py's own multi-rank paths on the one GPU a box offers: every line below is the code N ranks run -------------------
def _bench(argv, launcher, timeout=600):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable]
    if launcher:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    r = subprocess.run(cmd + [os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


QUICK = ["--steps", "2", "--warmup", "1", "--repeats", "1", "--size", "nyu", "--batch", "2", "--no-cpu-baseline", "--no-train-extra", "--no-nlspn-extra",
         "--no-head-extra", "--no-latency-b1", "--no-streams-extra"]


def test_bench_line_under_the_launcher_runs_its_collectives_on_rccl_with_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (how the driver starts every rank of an N-GPU run): process group on
    nccl = RCCL, barriers, the MAX all-reduce of the timed region, the rank census -- and the same line started plainly has no process group."""
    r, d = _bench(["--gpus", "1"] + QUICK, launcher=True)
    assert r.returncode == 0 and d is not None, r.stdout[-1500:] + r.stderr[-3000:]
    assert d["n_gpus"] == 1 and d["config"]["ranks_seen"] == 1 and d["config"]["parallelism"].startswith("dp1 ")
    assert d["config"]["process_group"] == "nccl (RCCL), world size 1" and "RCCL process group up: world size 1" in r.stderr
    assert d["config"]["global_batch"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    r2, d2 = _bench(["--gpus", "1"] + QUICK, launcher=False)
    assert r2.returncode == 0 and d2["config"]["process_group"] is None and d2["n_gpus"] == 1, r2.stderr[-2000:]


def test_asking_for_more_gpus_than_the_box_has_fails_at_once_with_the_stated_message():
    import time
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box runs the real thing")
    t0 = time.time()
    r, d = _bench(["--gpus", "2"] + QUICK, launcher=False, timeout=180)
    assert r.returncode != 0 and d is None and time.time() - t0 < 120
    assert "only 1 HIP device(s) visible" in r.stderr + r.stdout and "no CPU fallback" in r.stderr + r.stdout
    # ... and a launcher world size that contradicts --gpus is refused by every rank, not hung on
    env = {k: v for k, v in os.environ.items()}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29599")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + QUICK, env=env, capture_output=True, text=True, timeout=180, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr + r.stdout


def test_train_dp_under_the_launcher_takes_the_collective_path_with_one_rank():
    """`--mode train-dp` under the launcher on one GPU: the overlapped gradient reducer is ACTIVE (post-accumulate hooks, async all-reduce on RCCL's
    stream issued from inside backward, finish()) and SyncBatchNorm exchanges its statistics -- one rank, identity sums."""
    r, d = _bench(["--gpus", "1", "--mode", "train-dp", "--variant", "res", "--size", "nyu", "--batch", "2", "--steps", "2", "--warmup", "1"], launcher=True)
    assert r.returncode == 0 and d is not None, r.stdout[-1500:] + r.stderr[-3000:]
    c = d["config"]
    assert d["n_gpus"] == 1 and c["process_group"] == "nccl (RCCL), world size 1" and c["reducer_active"] and c["sync_batchnorm"]
    assert d["collectives_launched_in_backward"] >= 1 and d["backward_reads_kept_states"] and d["value"] > 0
    r2, d2 = _bench(["--gpus", "1", "--mode", "train-dp", "--variant", "res", "--size", "nyu", "--batch", "2", "--steps", "2", "--warmup", "1"], launcher=False)
    assert r2.returncode == 0 and not d2["config"]["reducer_active"] and d2["config"]["process_group"] is None


GRAPH_RULE_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.environ["DD_ROOT"])
if os.environ.get("TOUCH_HIP_FIRST") == "1":
    assert torch.cuda.is_available()              # a first HIP call BEFORE the package is imported: the runtime has read its switches
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
be = dda.HipDenoiser(torch.device("cuda", 0))
be.load_state_dict(synth.make_state_dict(7240)); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
inp = synth.make_inputs(3, 2, 9, 33)
x, c = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
outs = [be.denoise(x, c, 4, "f16r").cpu() for _ in range(3)]
assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
print("RULE", dda.GRAPH_REPLAY_SAFE, be.graph_replay, be.counter("graph_default"), be.counter("graph_launches") > 0, be.counter("eager_loops") > 0, float(outs[0].double().sum()))
"""


def test_the_binding_replays_graphs_only_where_the_fast_path_switch_was_in_the_environment():
    """Round 6: hipGraph replay only in a process whose environment had DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the package was imported (this session: tests/conftest.py);
    without it -- whether or not a HIP call came first -- the same kernels are enqueued eagerly and give the same numbers; DDEPTH_GRAPH overrides."""
    base = {k: v for k, v in os.environ.items() if k not in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "DDEPTH_GRAPH")}
    base["DD_ROOT"] = ROOT
    res = {}
    for tag, extra in (("exported", {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}), ("not_exported", {}), ("hip_first", {"TOUCH_HIP_FIRST": "1"}),
                       ("fast_path_on", {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"}), ("forced_off", {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0", "DDEPTH_GRAPH": "0"})):
        r = subprocess.run([sys.executable, "-c", GRAPH_RULE_SCRIPT], env=dict(base, **extra), capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RULE")]
        assert r.returncode == 0 and line, (tag, r.stdout[-1500:] + r.stderr[-2500:])
        res[tag] = line[-1].split()[1:]
    #                          GRAPH_REPLAY_SAFE  binding replays  C default  graph launches  eager loops
    assert res["exported"][:5] == ["True", "True", "1", "True", "True"] or res["exported"][:4] == ["True", "True", "1", "True"], res      # (a plan's first use is an eager pass in front of the capture)
    assert res["not_exported"][:2] == ["False", "False"] and res["not_exported"][3] == "False" and res["not_exported"][4] == "True", res
    assert res["hip_first"][:2] == ["False", "False"] and res["hip_first"][3] == "False", res
    assert res["fast_path_on"][:3] == ["False", "False", "0"] and res["fast_path_on"][3] == "False", res
    assert res["forced_off"][0] == "True" and res["forced_off"][1] == "False" and res["forced_off"][3] == "False", res
    assert len({r[5] for r in res.values()}) == 1, res          # the same numbers every way
