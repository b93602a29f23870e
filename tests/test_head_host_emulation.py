"""The drop-in head END TO END on the CPU: diffusiondepth_amd.head / modules / scheduler / backend (the product's own Python, incl. the
ctypes binding and the autograd Functions) driving the host-emulated build of the library (tests/host_emul: the library's own sources
compiled for x86), against the goldens minted from the reference's head class (tests/golden/make_golden.py).

Test infrastructure only.  The product binding refuses CPU tensors and loads only libddepth_hip.so; here the TEST patches its private
guards (tensor check, stream lookup, device scope) and hands it the emulated library, so that the host logic that otherwise first runs on
the GPU box -- parameter groups and their refresh (HipBound), RNG draw order, the 13-key output contract, loss.backward() through the loop
and the ddim_loss call -- is exercised in the CPU suite.  Under emulation "device" memory is host memory."""
from __future__ import annotations

import contextlib
import ctypes
import os

import numpy as np
import pytest
import torch

import diffusiondepth_amd as dda
from diffusiondepth_amd import backend as B_, modules as M_, synth
import hostemu_head

FULL = os.environ.get("DD_EMU_FULL") == "1"
CPU = torch.device("cpu")


@pytest.fixture(scope="module")
def lib():
    return hostemu_head.load()


@pytest.fixture()
def on_host(lib, monkeypatch):
    """Patch the binding's guards for the duration of one test; returns a factory of HipDenoiser objects bound to the emulated library."""
    make, made = hostemu_head.install(lib, monkeypatch.setattr)
    yield make
    hostemu_head.destroy(lib, made)


def _head(c, precision, T=None):
    sd = synth.make_state_dict(c["wseed"], "res", c["decoder_gain"], c["decoder_log_scale"])
    sd.update(synth.make_fpn_state_dict(c["fseed"]))
    head = dda.DDIMDepthEstimate_Res(in_channels=[64, 128, 256, 512], inference_steps=T or c["T"], num_train_timesteps=1000,
                                     depth_feature_dim=16, loss_cfgs=[], precision=precision)
    missing, unexpected = head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return head


def _inputs(c, grad=False):
    Bn, H, W = c["B"], c["H"], c["W"]
    fp = [torch.from_numpy(f).requires_grad_(grad) for f in synth.make_backbone_features(c["iseed"], Bn, H, W)]
    gt = torch.from_numpy(synth.make_gt_depth(c["iseed"] + 1, Bn, H, W))
    h, w = synth.latent_hw(H, W)
    return fp, gt, synth.make_inputs(c["iseed"] + 2, Bn, h, w)


@contextlib.contextmanager
def _draws(inp):
    """The reference's RNG draws, injected in its order: device randn for x_T (...res.py:277), CPU randn for the loss noise (:203), randint (:207)."""
    draws = [torch.from_numpy(inp["x_T"]), torch.from_numpy(inp["noise"])]
    real_randn, real_randint = torch.randn, torch.randint
    torch.randn = lambda *a, **k: draws.pop(0)
    torch.randint = lambda *a, **k: torch.from_numpy(inp["timesteps"])
    try:
        yield
    finally:
        torch.randn, torch.randint = real_randn, real_randint


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_vis_head_forward_vs_reference_golden(on_host, golden, cases, monkeypatch):
    """DDIMDepthEstimate_ResVis.forward in eval mode (encoder, condition FPN by dd_condition, 5-step loop through dd_denoise_trace, decoder of every
    intermediate sample, ddim_loss through dd_add_noise-less q_sample + dd_denoise_once) against the reference head's outputs.
    16-bit operands keep the emulation short: the bound is the f16 class, the fp32 gate (1e-3 abs on depth) runs with DD_EMU_FULL=1."""
    c, g = cases["head_res_vis"], golden("head_res_vis")
    prec = "fp32" if FULL else "f16"
    head = dda.DDIMDepthEstimate_ResVis(in_channels=[64, 128, 256, 512], inference_steps=c["T"], precision=prec).eval()
    sd = synth.make_state_dict(c["wseed"], "res", c["decoder_gain"], c["decoder_log_scale"])
    sd.update(synth.make_fpn_state_dict(c["fseed"]))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    fp, gt, inp = _inputs(c)
    monkeypatch.setattr(type(head), "_on_hip", staticmethod(lambda tensors: True))       # condition FPN in the library too (dd_condition)
    seen = []
    real_cond_arg = B_.HipDenoiser._cond_arg
    monkeypatch.setattr(B_.HipDenoiser, "_cond_arg", lambda self, cond, precision: (seen.append(real_cond_arg(self, cond, precision)), seen[-1])[1])
    with _draws(inp), torch.no_grad():
        out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
    assert set(out) == set(cases["head_res"]["output_keys"])
    be = head._bound.backend
    assert be is not None and be._lib.emu_launch_count() > 0, "the library did not run"
    assert seen == [None, None], "the loop and the ddim_loss call must pick up the map dd_condition left in the handle (no re-conversion)"
    tol = 1e-3 if FULL else 2e-2 * float(np.abs(g["pred"]).max())
    assert float(np.abs(out["pred"].numpy() - g["pred"]).max()) < tol
    assert len(out["pred_inter"]) == c["T"]
    for k in range(c["T"]):                                           # every intermediate sample, decoded (...res_vis.py:141-143)
        assert float(np.abs(out["pred_inter"][k].numpy() - g["pred_inter"][k]).max()) < tol * (1.0 if FULL else max(1.0, float(np.abs(g["pred_inter"][k]).max()) / float(np.abs(g["pred"]).max())))
    assert torch.equal(out["pred_init"], out["gt_map_t"]) and out["pred_init"].shape == (c["B"], 16, c["H"] // 2, c["W"] // 2)
    gl = float(g["ddim_loss"][0])
    assert abs(float(out["ddim_loss"]) - gl) < (1e-4 if FULL else 2e-2) * max(1.0, abs(gl))


def test_training_step_with_optimizer_refresh_vs_reference_golden(on_host, golden, cases, monkeypatch):
    """One .train() step of DDIMDepthEstimate_Res (loop + ddim_loss forward AND backward in the library through the autograd Functions,
    BatchNorm FPN / codec in torch) against the reference head under autograd (head_train_res.npz), then optimizer.step() and a second
    forward: the denoiser group -- and only it -- is refreshed, here through dd_set_weight_device."""
    monkeypatch.setenv("DDEPTH_DEVICE_WEIGHTS", "1")
    c, g = cases["head_train_res"], golden("head_train_res")
    head = _head(c, "naive_fp32" if not FULL else "fp32").train()
    fp, gt, inp = _inputs(c, grad=True)
    with _draws(inp):
        out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
        loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
        loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"][0])) <= 1e-4 * abs(float(g["loss"][0]))
    errs = {"pred": _rel(out["pred"].detach().numpy(), g["pred"]), "grad_fp3": _rel(fp[3].grad.numpy(), g["grad_fp3"]),
            "grad_fp0": _rel(fp[0].grad.numpy()[:, :4], g["grad_fp0_ch0_4"])}
    named = dict(head.named_parameters())
    for k in c["grad_keys"]:
        got = named[k].grad.numpy()
        if got.size > 5000:
            got = got.reshape(-1)[::c["grad_stride"]]
        errs[k] = _rel(got, g["grad." + k])
    bad = {k: v for k, v in errs.items() if v > 1e-2}
    assert not bad, bad
    # optimizer step -> the next forward refreshes the denoiser group only (codec and FPN run in torch while training)
    be = head._bound.backend
    uploads = []
    real = be.load_state_dict
    monkeypatch.setattr(be, "load_state_dict", lambda sd, **k: (uploads.append(sorted({n.split(".")[0] for n in sd})), real(sd, **k))[1])
    d0 = be.weights_digest()
    torch.optim.SGD(head.parameters(), lr=1e-2).step()
    with _draws(inp):
        out2 = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
    assert uploads == [["model"]]
    assert be.weights_digest() != d0
    fresh = on_host(CPU)
    fresh.load_state_dict({"model." + k: v for k, v in head.model.state_dict().items()}, device_route=False)
    assert fresh.weights_digest() == be.weights_digest()           # device route == host route on the updated values
    assert not torch.equal(out2["pred"], out["pred"])


@pytest.mark.skipif(not FULL, reason="T = 20 on the UpSample_add denoiser: minutes of emulation, set DD_EMU_FULL=1")
@pytest.mark.parametrize("case,cls,chans", [("head_mpvit_hahi", "DDIMDepthEstimate_MPVIT_ADDHAHI", (128, 216, 288, 288)),
                                            ("head_swin_hahi", "DDIMDepthEstimate_Swin_ADDHAHI", (192, 384, 768, 1536))], ids=["mpvit", "swin"])
def test_hahi_head_forward_vs_reference_golden(on_host, golden, cases, monkeypatch, case, cls, chans):
    """The reference's headline head (HAHI neck in torch -> dd_condition at the Swin-L / MPViT-small widths -> bilinear upsample +
    UpSample_add denoiser, 20 steps -> decoder -> ddim_loss), f16 operands, against the reference head's outputs."""
    c, g = cases[case], golden(case)
    sd = synth.make_state_dict(c["wseed"], "swin", c["decoder_gain"], c["decoder_log_scale"])
    sd.update({k: v for k, v in synth.make_fpn_state_dict(c["fseed"], in_channels=chans).items() if not k.startswith("convup_fp")})
    sd.update(synth.make_hahi_state_dict(c["hseed"], chans))
    head = getattr(dda, cls)(in_channels=list(chans), inference_steps=c["T"], num_train_timesteps=1000, depth_feature_dim=16, loss_cfgs=[],
                             precision="f16", neck_autocast=False).eval()
    missing, unexpected = head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    monkeypatch.setattr(type(head), "_on_hip", staticmethod(lambda tensors: True))
    Bn, H, W = c["B"], c["H"], c["W"]
    fp = [torch.from_numpy(f) for f in synth.make_backbone_features(c["iseed"], Bn, H // 2, W // 2, in_channels=chans)]
    gt = torch.from_numpy(synth.make_gt_depth(c["iseed"] + 1, Bn, H, W))
    h, w = synth.latent_hw(H, W)
    inp = synth.make_inputs(c["iseed"] + 2, Bn, h, w, (fp[0].shape[2], fp[0].shape[3]))
    with _draws(inp), torch.no_grad():
        out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
    assert set(out) == set(cases["head_res"]["output_keys"]) and out["pred_inter"] is None
    scale = float(np.abs(g["pred"]).max())
    assert float(np.abs(out["pred"].numpy() - g["pred"]).max()) < 2e-2 * scale
    assert float(np.abs(out["pred_init"].numpy() - g["pred_init"]).max()) < 2e-5
    gl = float(g["ddim_loss"][0])
    assert abs(float(out["ddim_loss"]) - gl) < 2e-2 * max(1.0, abs(gl))


def test_model_facade_on_the_plumbing_configuration(on_host, monkeypatch):
    """BASELINE.json configs[0] -- "ResNet-18 backbone + 64x64 latent, 5-step DDIM, batch=1 on CPU (plumbing)": the model facade's
    forward(sample) -> dict (reference src/model/diffusion_dcbase_model.py:186-224) with mmbev_res18 in torch and the whole head in the
    (emulated) library, f16 operands, against the fp64 oracle run on the same condition map and the same x_T."""
    from diffusiondepth_amd import model as MD
    from oracle import ddim_oracle as O
    torch.manual_seed(7240)                                                       # reference default seed (src/config.py:44)
    net = MD.Diffusion_DCbase_Model(MD.default_args(backbone_name="mmbev_res18", inference_steps=5, precision="f16")).eval()
    head = net.depth_head
    monkeypatch.setattr(type(head), "_on_hip", staticmethod(lambda tensors: True))
    rs = np.random.RandomState(5)
    rgb = torch.from_numpy(rs.standard_normal((1, 3, 128, 128)).astype(np.float32))
    gt = torch.from_numpy(synth.make_gt_depth(6, 1, 128, 128))
    inp = synth.make_inputs(7, 1, 64, 64)
    with _draws(inp), torch.no_grad():
        out = net({"rgb": rgb, "gt": gt, "dep": gt, "depth_map": gt, "depth_mask": gt > 0})
    assert out["pred"].shape == (1, 1, 128, 128) and out["pred_init"].shape == (1, 16, 64, 64)
    assert torch.isfinite(out["pred"]).all() and torch.isfinite(out["ddim_loss"])
    assert head._bound.backend.counter("graph_launches") >= 1, "the 5-step loop did not run as one (recorded) graph"
    # the same computation restated: torch backbone + torch FPN -> oracle loop + decoder (fp64) on the same x_T
    with torch.no_grad():
        fp = net.depth_backbone(rgb)
        monkeypatch.setattr(head, "_hip_fpn", False)
        cond = head.aggregate_condition(fp).numpy()
    sd = {k: v.numpy() for k, v in head.state_dict().items()}
    lat = O.ddim_loop(sd, inp["x_T"], cond, 5)
    want = O.decode(sd, lat)
    got = out["pred"].numpy()
    assert float(np.abs(got - want).max()) < 2e-2 * max(float(np.abs(want).max()), 1e-3), (float(np.abs(got - want).max()), float(np.abs(want).max()))


# ---- data-parallel training, two ranks over gloo: the one exchange step of the path end to end ---------------------------------------
def _dp_step(head, c, rank_items, reducer=None):
    """One training step on the images ``rank_items`` of the head_train_res case (B = len(rank_items)); returns {name: grad}."""
    fp_all, gt_all, inp = _inputs(c, grad=False)
    idx = torch.tensor(rank_items)
    fp = [f[idx].clone() for f in fp_all]
    gt = gt_all[idx].clone()
    sub = {k: (v[rank_items] if isinstance(v, np.ndarray) and v.shape[:1] == (c["B"],) else v) for k, v in inp.items()}
    for p in head.parameters():
        p.grad = None
    with _draws(sub):
        out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
        loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
        loss.backward()
    if reducer is not None:
        reducer.finish()
    return {k: (p.grad.clone() if p.grad is not None else None) for k, p in head.named_parameters()}


def _dp_worker(rank, world, port, out, case, sync_bn=False):
    import torch.distributed as dist
    from diffusiondepth_amd import dist as ddist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    ddist.init_from_env("gloo")
    lib = hostemu_head.load()
    hostemu_head.install(lib, setattr)
    head = _head(case, "naive_fp32").train()
    if sync_bn:
        head = ddist.convert_sync_batchnorm(head)          # reference src/main.py:128 (apex.parallel.convert_syncbn_model)
    red = ddist.OverlappedGradReducer(list(head.parameters()), bucket_bytes=1 << 20)     # several buckets, launched from the autograd hooks
    grads = _dp_step(head, case, ddist.shard_indices(case["B"], rank, world), red)
    red.close()
    if rank == 0:
        torch.save({k: v for k, v in grads.items() if v is not None}, out)
    dist.barrier()
    dist.destroy_process_group()


# (the per-rank-statistics variant costs a minute of emulation: with DD_EMU_FULL=1 only; SyncBatchNorm -- what bench.py --mode train-dp and the
#  reference run -- stays in the default set)
@pytest.mark.parametrize("sync_bn", ([False, True] if FULL else [True]), ids=(["bn-per-rank", "sync-bn"] if FULL else ["sync-bn"]))
def test_two_rank_data_parallel_training_step_equals_the_average_of_its_shards(on_host, cases, tmp_path, sync_bn):
    """Two processes (gloo; RCCL on the GPU box), one image of the head_train_res case each: forward + backward through the library,
    gradient buckets all-reduced from autograd hooks while backward runs (dist.OverlappedGradReducer, replaces apex DDP:
    src/main.py:106-114,148).  What rank 0 ends with must equal the average of the two shards' gradients computed in ONE process
    (BatchNorm statistics per rank, as without SyncBN) -- for the denoiser parameters, whose gradients come out of dd_denoise_backward /
    dd_denoise_once_backward, and for the torch-side FPN / codec.  With dist.convert_sync_batchnorm (what the reference does to every
    BatchNorm before DDP) the two ranks must instead equal ONE process on the whole batch: the data-parallel step is then exactly the
    large-batch step."""
    import socket
    import torch.multiprocessing as mp
    c = dict(cases["head_train_res"])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp_grads.pt")
    mp.spawn(_dp_worker, args=(2, port, out, c, sync_bn), nprocs=2, join=True)
    got = torch.load(out)
    head = _head(c, "naive_fp32").train()
    if sync_bn:
        g0 = g1 = _dp_step(head, c, [0, 1])
    else:
        g0, g1 = _dp_step(head, c, [0]), _dp_step(head, c, [1])
    checked = 0
    for k in ("model.pred.0.weight", "model.noise_embedding.3.weight", "model.pred.3.bias", "model.noise_embedding.1.weight", "model.time_embedding.weight",
              "conv_lateral.0.0.weight", "conv_up.1.0.weight", "depth_transform.conv_inv_transform.0.weight"):
        want = 0.5 * (g0[k] + g1[k])
        # round-off class: the workers' torch ops (BatchNorm FPN / codec) run with another thread partition than this process, and a
        # 1e-7 input difference can flip a ReLU mask bit (the GPU training test's tolerance, for the same reason, is 1e-2); a missing
        # average or an unreduced bucket would be an O(1) error
        assert float((got[k] - want).abs().max()) <= 2e-3 * float(want.abs().max()), k
        assert float(want.abs().max()) > 0
        checked += 1
    assert checked == 8


def test_the_host_never_blocks_in_steady_state(on_host, lib, monkeypatch):
    """Runtime calls after which the HOST waits for the device (hipDeviceSynchronize / hipStreamSynchronize / hipEventSynchronize, blocking
    hipMemcpy, hipMalloc / hipFree; counted by the emulated runtime): once plans, graphs and workspaces exist, an eval forward of the head
    -- encoder, dd_condition, the loop, decoder, the ddim_loss call -- makes NONE, nor does a training step's backward; the per-step
    parameter refresh costs 2 by the device route (one device sync before the images are overwritten, one stream sync after) against dozens
    by the host route (a synchronous upload per packed image)."""
    lib.emu_blocking_calls.restype = ctypes.c_ulong
    sd = synth.make_state_dict(7240)
    sd.update(synth.make_fpn_state_dict(7241))
    Bn, H, W = 1, 16, 48
    fp = [torch.from_numpy(f) for f in synth.make_backbone_features(1, Bn, H, W)]
    gt = torch.from_numpy(synth.make_gt_depth(2, Bn, H, W))
    head = dda.DDIMDepthEstimate_Res(precision="f16", inference_steps=3, loss_noise_device="device").eval()
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    monkeypatch.setattr(type(head), "_on_hip", staticmethod(lambda tensors: True))
    with torch.no_grad():
        head(fp, gt, gt > 0, gt_depth_map=gt)                                   # first call: uploads, plans, graph capture
        n0 = lib.emu_blocking_calls()
        head(fp, gt, gt > 0, gt_depth_map=gt)
    assert lib.emu_blocking_calls() == n0
    train = dda.DDIMDepthEstimate_Res(precision="f16", inference_steps=2).train()
    train.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    opt = torch.optim.SGD(train.parameters(), lr=1e-4)
    per_route = {}
    for route in ("0", "1"):
        monkeypatch.setenv("DDEPTH_DEVICE_WEIGHTS", route)
        for it in range(2):
            fpg = [f.clone().requires_grad_(True) for f in fp]
            c0 = lib.emu_blocking_calls()
            out = train(fpg, gt, gt > 0, gt_depth_map=gt)
            c1 = lib.emu_blocking_calls()
            ((out["pred"] - gt).abs().mean() + out["ddim_loss"]).backward()
            c2 = lib.emu_blocking_calls()
            opt.step()
            opt.zero_grad()
        per_route[route] = (c1 - c0, c2 - c1)                                   # second iteration = steady state
    assert per_route["1"] == (2, 0), per_route
    assert per_route["0"][1] == 0 and per_route["0"][0] > 20, per_route


def test_training_refuses_to_continue_on_non_finite_gradients(on_host, monkeypatch):
    """The always-on guard of the training path (modules._note_grads / HipBound.check_grad_guard; ADVICE r5: round 4's library trained on NaN
    parameters silently): a backward of the library that returns a NaN makes the NEXT training forward raise before it uploads parameters; a clean
    step passes, and DDEPTH_GRAD_GUARD=0 (module switch GRAD_GUARD) restores the old behaviour."""
    from diffusiondepth_amd import modules as M
    sd = synth.make_state_dict(7240)
    sd.update(synth.make_fpn_state_dict(7241))
    Bn, H, W = 1, 16, 48
    fp = [torch.from_numpy(f) for f in synth.make_backbone_features(1, Bn, H, W)]
    gt = torch.from_numpy(synth.make_gt_depth(2, Bn, H, W))
    head = dda.DDIMDepthEstimate_Res(precision="naive_fp32", inference_steps=2).train()
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    monkeypatch.setattr(type(head), "_on_hip", staticmethod(lambda tensors: True))

    def step(poison):
        out = head([f.clone() for f in fp], gt, gt > 0, gt_depth_map=gt)
        loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
        (loss * float("nan") if poison else loss).backward()
        head.zero_grad()
    step(False)
    step(False)                     # a clean step's flag is consumed by the next forward without complaint
    step(True)                      # NaN gradients come back from dd_denoise_backward / dd_denoise_once_backward ...
    with pytest.raises(FloatingPointError, match="non-finite gradients"):
        step(False)                 # ... and the next forward refuses to refresh the parameters
    step(False)                     # (the flag is consumed by the raise: training can be resumed by whoever handles it)
    monkeypatch.setattr(M, "GRAD_GUARD", False)
    step(True)
    step(False)


def test_loop_backward_reads_the_states_the_forward_kept(on_host):
    """Training: dd_denoise with option keep_trajectory leaves the state entering every step behind and hands out a ticket;
    dd_denoise_backward given that ticket skips its second forward loop (counter trajectory_reuses) -- and, the activations of every step
    having been kept too (default), every per-step recompute -- and returns what the regenerating path returns.  A ticket whose states were overwritten by a later forward, or that predates a parameter update, is refused (regenerated)."""
    be = on_host(CPU)
    sd = synth.make_state_dict(7240)
    be.load_state_dict(sd)
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    T, prec = 2, ("fp32" if FULL else "f16")
    be.set_option("hoist_cond", 0)                 # without the hoisted condition term the kept and the regenerated states are the same bytes
    inp, inp2 = synth.make_inputs(5, 1, 6, 33), synth.make_inputs(6, 1, 6, 33)
    x, cond, x2 = torch.from_numpy(inp["x_T"]), torch.from_numpy(inp["cond"]), torch.from_numpy(inp2["x_T"])
    g = torch.from_numpy(np.random.RandomState(3).standard_normal(inp["x_T"].shape).astype(np.float32))
    names = ["model.pred.3.weight", "model.noise_embedding.0.weight", "model.pred.1.bias", "model.time_embedding.weight"]

    def backward(ticket):
        be.zero_grad()
        gx, gc = be.denoise_backward(x, cond, g, T, prec, need_grad_xT=True, trajectory_ticket=ticket)
        return [gx, gc] + [be.grad(n) for n in names]

    def same(a, b):
        return all(float((p - q).abs().max()) <= 1e-6 * float(q.abs().max()) for p, q in zip(a, b))    # fp64 atomics: summation order

    x0 = be.denoise(x, cond, T, prec)
    assert be.last_trajectory_ticket == 0
    want = backward(0)
    assert be.counter("trajectory_reuses") == 0
    x0k = be.denoise(x, cond, T, prec, keep_trajectory=True)
    tk = be.last_trajectory_ticket
    assert tk > 0 and torch.equal(x0k, x0)
    assert same(backward(tk), want) and be.counter("trajectory_reuses") == 1          # (a second backward on the same ticket reuses again)
    be.denoise(x2, cond, T, prec, keep_trajectory=True)                                # the plan's states are now another call's
    assert be.last_trajectory_ticket == tk + 1
    assert same(backward(tk), want) and be.counter("trajectory_reuses") == 1
    # the same with the states only (activation budget 0: every step's forward is recomputed from its kept state)
    be.set_option("keep_activations_mb", 0)
    plans = be.counter("plans")
    assert torch.equal(be.denoise(x, cond, T, prec, keep_trajectory=True), x0) and be.counter("plans") == plans + 1
    assert same(backward(be.last_trajectory_ticket), want) and be.counter("trajectory_reuses") == 2
    be.set_option("keep_activations_mb", 65536)
    be.denoise(x, cond, T, prec, keep_trajectory=True)
    tk3 = be.last_trajectory_ticket
    # the single call (ddim_loss): the forward's activations stay in its plan; its backward with the ticket recomputes nothing
    t = torch.tensor([400])
    def once(ticket):
        be.zero_grad()
        gx1, gc1 = be.denoise_once_backward(x, t, cond, g, prec, trajectory_ticket=ticket)
        return [gx1, gc1] + [be.grad(n) for n in names]
    eps0 = be.denoise_once(x, t, cond, prec)
    want1 = once(0)
    n_re = be.counter("trajectory_reuses")
    eps1 = be.denoise_once(x, t, cond, prec, keep_trajectory=True)
    tk1 = be.last_trajectory_ticket
    assert tk1 > 0 and torch.equal(eps1, eps0)
    assert same(once(tk1), want1) and be.counter("trajectory_reuses") == n_re + 1
    be.denoise_once(x2, t, cond, prec, keep_trajectory=True)                             # overwritten by another call: the old ticket is refused
    assert same(once(tk1), want1) and be.counter("trajectory_reuses") == n_re + 1
    bumped = {"model.pred.4.weight": sd["model.pred.4.weight"] * 1.5}                   # the last GroupNorm's gamma
    be.load_state_dict(bumped)                                                          # parameters changed after the forward
    got = backward(tk3)
    assert be.counter("trajectory_reuses") == 3
    assert not same(got, want)                                                          # ... and the gradient is the new parameters' one


def test_kept_activation_budget_is_one_figure_for_the_handle(on_host, monkeypatch):
    """the per-step activation slots a training forward keeps (PlanKey::keep == 2) are budgeted ONCE per handle (option
    keep_activations_mb) and against the free device memory; a plan whose trajectory has been consumed by its backward (or invalidated) is
    dropped when another shape needs the room; when nothing can be dropped the forward keeps the states only (keep == 1) and the backward
    recomputes -- with the same gradients."""
    be = on_host(CPU)
    be.load_state_dict(synth.make_state_dict(7240))
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    T, prec = 2, "f16"
    be.set_option("hoist_cond", 0)
    shapes = [(1, 6, 33), (1, 8, 33)]
    ins = [synth.make_inputs(5 + i, *sh) for i, sh in enumerate(shapes)]
    g = [torch.from_numpy(np.random.RandomState(3).standard_normal(i["x_T"].shape).astype(np.float32)) for i in ins]
    per_step = [B * h * w * ((64 + 256 + 64) * 2 + 16 * 4) for (B, h, w) in shapes]          # f16 y1..y3 + fp32 y4
    assert max(per_step) * T < (1 << 20)

    def fwd_bwd(i):
        x, c = torch.from_numpy(ins[i]["x_T"]), torch.from_numpy(ins[i]["cond"])
        be.zero_grad()
        be.denoise(x, c, T, prec, keep_trajectory=True)
        gx, _ = be.denoise_backward(x, c, g[i], T, prec, need_grad_xT=True, trajectory_ticket=be.last_trajectory_ticket)
        return gx, be.grad("model.pred.3.weight").clone()

    be.set_option("keep_activations_mb", 1)                    # room for ONE of the two shapes' slots... (each < 1 MiB, together > 1 MiB?)
    want = [fwd_bwd(0), fwd_bwd(1)]
    # both fit under 1 MiB only if their sum does: force the squeeze through the free-memory reading instead
    monkeypatch.setenv("HOSTEMU_FREE_BYTES", str(int(max(per_step) * T * 1.2)))
    r0 = be.counter("trajectory_reuses")
    got0 = fwd_bwd(0)            # existing keep-2 plan: reused as it is
    got1 = fwd_bwd(1)
    assert be.counter("trajectory_reuses") == r0 + 2
    for (a, b), (p, q) in zip(want, [got0, got1]):
        assert float((a - p).abs().max()) <= 1e-6 * float(a.abs().max()) and float((b - q).abs().max()) <= 1e-6 * float(b.abs().max())
    # a THIRD shape under the squeeze: a consumed plan is dropped for it (the plan count does not grow by a keep-2 plan on top of two)
    inp3 = synth.make_inputs(9, 1, 7, 33)
    x3, c3 = torch.from_numpy(inp3["x_T"]), torch.from_numpy(inp3["cond"])
    plans = be.counter("plans")
    be.denoise(x3, c3, T, prec, keep_trajectory=True)
    assert be.counter("plans") <= plans                      # one stale keep-2 plan went, the new one came
    # nothing droppable (the trajectory just kept is live) and no room: states only, the backward recomputes, same gradient as the naive order
    monkeypatch.setenv("HOSTEMU_FREE_BYTES", "1024")
    x, c = torch.from_numpy(ins[1]["x_T"]), torch.from_numpy(ins[1]["cond"])
    inp4 = synth.make_inputs(11, 1, 5, 33)
    x4, c4 = torch.from_numpy(inp4["x_T"]), torch.from_numpy(inp4["cond"])
    g4 = torch.from_numpy(np.random.RandomState(4).standard_normal(inp4["x_T"].shape).astype(np.float32))
    be.zero_grad()
    be.denoise(x4, c4, T, prec, keep_trajectory=True)        # keep == 1
    gx_k, _ = be.denoise_backward(x4, c4, g4, T, prec, need_grad_xT=True, trajectory_ticket=be.last_trajectory_ticket)
    be.zero_grad()
    gx_n, _ = be.denoise_backward(x4, c4, g4, T, prec, need_grad_xT=True)
    assert float((gx_k - gx_n).abs().max()) <= 1e-6 * float(gx_n.abs().max())


@pytest.mark.parametrize("shape", [((2, 7, 19), (4, 10)), ((1, 4, 141), (2, 70))] + ([((1, 9, 33), (3, 5)), ((1, 6, 40), (6, 40))] if FULL else []),
                         ids=["x2", "two-segments"] + (["x4-ragged", "same-size"] if FULL else []))
def test_swin_condition_gradient_tiled_adjoint_equals_the_plain_one(on_host, shape):
    """Swin variant, 16-bit backward: dLoss/dcond passes through the adjoint of the align_corners bilinear upsampling.  The tiled separable
    kernel (dd_bwd.hip: upsample_adjoint_tiled_kernel) must return what the one-thread-per-piece kernel returns (same fp32 weights, other
    summation order), at the condition map's own size, over several source-column segments and with accumulation over two loop steps."""
    (B, h, w), (ch, cw) = shape
    be = on_host(CPU, "swin")
    be.load_state_dict(synth.make_state_dict(7301, "swin"))
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    inp = synth.make_inputs(11, B, h, w, (ch, cw))
    x, cond = torch.from_numpy(inp["x_T"]), torch.from_numpy(inp["cond"])
    g = torch.from_numpy(np.random.RandomState(5).standard_normal(inp["x_T"].shape).astype(np.float32))
    out = {}
    for tiled in (0, 1):
        be.set_option("adjoint_tiled", tiled)
        be.zero_grad()
        _, gc1 = be.denoise_once_backward(x, torch.full((B,), 300), cond, g, "bf16")
        gc2 = be.denoise_backward(x, cond, g, 2, "bf16")[1] if (B > 1 or FULL) else gc1      # two steps: the second one accumulates
        out[tiled] = (gc1, gc2)
    for a, b in zip(out[1], out[0]):
        assert a.shape == (B, 256, ch, cw) and float(b.abs().max()) > 0
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


@pytest.mark.parametrize("variant,hw,cond_hw", [("res", (9, 21), None)] + ([("swin", (7, 19), (4, 10))] if FULL else []))
def test_16bit_backward_close_to_the_fp32_backward(on_host, variant, hw, cond_hw):
    """The vectorised GroupNorm-backward kernels of the 16-bit modes (gn_bwd_reduce_blocked / gn_bwd_apply_blocked: two pixels per trip,
    ragged last pair, slabs) against the generic fp32 path of the same library on a pixel count that is no multiple of anything.  The
    distance is dominated by ReLU-mask flips, not by rounding (a pre-activation within 16-bit distance of zero changes its whole gradient
    contribution; measured on these inputs: relative L2 0.05-0.08 in f16, 0.07-0.10 in bf16, identical to four digits before and after the
    kernels were vectorised further), so the gate only catches structural errors -- a wrong pixel, a dropped pair -- which show up as O(1)."""
    be = on_host(CPU, variant)
    be.load_state_dict(synth.make_state_dict(7240, variant))
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    B, (h, w) = 2, hw
    inp = synth.make_inputs(17, B, h, w, cond_hw) if cond_hw else synth.make_inputs(17, B, h, w)
    x, cond = torch.from_numpy(inp["x_T"]), torch.from_numpy(inp["cond"])
    g = torch.from_numpy(np.random.RandomState(9).standard_normal(inp["x_T"].shape).astype(np.float32))
    t = torch.tensor([120, 700])
    names = list(be.param_shapes())
    res = {}
    for prec in ("fp32", "bf16", "f16"):
        be.zero_grad()
        gx, gc = be.denoise_once_backward(x, t, cond, g, prec)
        res[prec] = [gx, gc] + [be.grad(n) for n in names]
    for prec, tol in (("bf16", 0.15), ("f16", 0.12)):
        bad = {}
        for k, a, b in zip(["grad_x", "grad_cond"] + names, res[prec], res["fp32"]):
            if k == "model.time_embedding.weight":
                a, b = a[t], b[t]
            e = float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
            if e > tol:
                bad[k] = e
        assert not bad, (prec, bad)


@pytest.mark.parametrize("variant,cond_hw,precs,lanes", [("res", None, ("bf16",), (2, 3) if FULL else (2,))] + ([("swin", (3, 7), ("fp32",), (2,))] if FULL else []),
                         ids=["res-bf16"] + (["swin-fp32"] if FULL else []))
def test_concurrent_lanes_return_the_single_stream_result(on_host, variant, cond_hw, precs, lanes):
    """dd_set_option("streams", S): dd_denoise runs a batch as S concurrent sub-batches (own plans, buffers and graphs per lane; fork / join
    by events on the caller's stream).  Same bytes as one stream -- the images are independent -- with an explicit condition tensor (each
    lane converts its slice; Swin: upsamples it), uneven split (3 images on 2 lanes), replayed lane graphs; then the training pair
    (state-keeping forward + backward) as lanes.  (The lanes' slices of a condition map left by dd_condition and the Swin variant on the
    GPU: tests/test_gpu_fpn.py, tests/test_gpu_backward.py; Swin under emulation with DD_EMU_FULL=1.)"""
    be = on_host(CPU, variant)
    be.load_state_dict(synth.make_state_dict(7240, variant))
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    B, h, w, T = (3, 7, 19, 1) if variant == "res" else (2, 5, 13, 1)        # (the Swin denoiser is 5x the work per pixel)
    inp = synth.make_inputs(23, B, h, w, cond_hw) if cond_hw else synth.make_inputs(23, B, h, w)
    x, cond = torch.from_numpy(inp["x_T"]), torch.from_numpy(inp["cond"])
    for prec in precs:
        be.set_option("streams", 1)
        want = be.denoise(x, cond, T, prec)
        for S in lanes:
            be.set_option("streams", S)
            n0 = be.counter("lane_calls")
            got = be.denoise(x, cond, T, prec)
            assert be.counter("lane_calls") == n0 + 1
            assert torch.equal(got, want), (prec, S)
        assert torch.equal(be.denoise(x, cond, T, prec), want)               # second call: the lanes' graphs replay
    be.set_option("streams", 1)
    if variant != "res":
        return
    # training: forward that keeps its states / activations + backward, as lanes vs on one stream.  Every lane keeps its images in its own
    # plan under ONE ticket, the backward splits alike, parameter gradients are summed over the lanes' gradient sets at the join.
    prec = precs[0]
    g = torch.from_numpy(np.random.RandomState(7).standard_normal(inp["x_T"].shape).astype(np.float32))
    names = ["model.pred.3.weight", "model.noise_embedding.3.weight", "model.pred.1.weight", "model.pred.0.bias", "model.time_embedding.weight"]
    res = {}
    for S in (1, 2):
        be.set_option("streams", S)
        n0, r0 = be.counter("lane_calls"), be.counter("trajectory_reuses")
        out = be.denoise(x, cond, 2, prec, keep_trajectory=True)
        be.zero_grad()
        gx, gc = be.denoise_backward(x, cond, g, 2, prec, need_grad_xT=True, trajectory_ticket=be.last_trajectory_ticket)
        assert be.counter("trajectory_reuses") == r0 + 1 and be.counter("lane_calls") == n0 + (2 if S > 1 else 0)
        res[S] = [out, gx, gc] + [be.grad(n) for n in names]
    be.set_option("streams", 1)
    assert torch.equal(res[2][0], res[1][0]) and torch.equal(res[2][1], res[1][1]) and torch.equal(res[2][2], res[1][2])     # per-image results
    for a, b_ in zip(res[2][3:], res[1][3:]):                                                                                # sums over the batch: order
        assert float((a - b_).abs().max()) <= 2e-5 * float(b_.abs().max())


@pytest.mark.parametrize("cls,chans", [("DDIMDepthEstimate_MPVIT_ADDHAHI", (128, 216, 288, 288))] + ([("DDIMDepthEstimate_Swin_ADDHAHI", (192, 384, 768, 1536))] if FULL else []),
                         ids=["mpvit"] + (["swin"] if FULL else []))
def test_neck_in_the_library_equals_the_pytorch_neck(on_host, monkeypatch, cls, chans):
    """dd_neck_condition (the HAHI neck's twelve convolutions + the FPN, all in the library) against the PyTorch neck feeding the library's
    FPN, fp32 kernels, on a small odd-sized pyramid.  MPViT widths: 216 channels carried as 224 (zero weights, a gap in the fusion
    convolution's input channels), couts rounded up to whole 64-cout workgroup tiles whose padding is never stored."""
    sd = synth.make_state_dict(7240, "swin")
    sd.update({k: v for k, v in synth.make_fpn_state_dict(7241, in_channels=chans).items() if not k.startswith("convup_fp")})
    sd.update(synth.make_hahi_state_dict(7242, chans))
    head = getattr(dda, cls)(in_channels=list(chans), inference_steps=2, num_train_timesteps=1000, depth_feature_dim=16, loss_cfgs=[],
                             precision="fp32", neck_autocast=False).eval()
    missing, unexpected = head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    monkeypatch.setattr(type(head), "_on_hip", staticmethod(lambda tensors: True))
    fp = [torch.from_numpy(f) for f in synth.make_backbone_features(3, 1, 36, 52, in_channels=chans)]      # 9x13, 5x7, 3x4, 2x2: both pool size fixes
    assert head._hip_neck
    with torch.no_grad():
        c_lib = head.aggregate_condition(fp, neck_in_library=True)
        n0 = head._bound.backend.counter("neck_launches")
        c_ref = head.aggregate_condition([f.float() for f in head.hahineck(fp)])
    assert n0 == 12 and head._bound.backend.counter("neck_launches") == 12
    assert c_lib.shape == c_ref.shape == (1, 256, fp[0].shape[2], fp[0].shape[3])
    assert float((c_lib - c_ref).abs().max()) <= 1e-4 * max(1.0, float(c_ref.abs().max()))
