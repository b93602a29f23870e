"""GPU (`-m gpu`): the HIP path, called through the C ABI, against
  (1) the golden vectors minted from the reference's own classes (tests/golden/*.npz),
  (2) the fp64 NumPy oracle on seeded inputs (ragged / odd sizes, batch > 1),
  (3) the torch-CPU port at the full NYU / KITTI sizes of BASELINE.json,
  (4) size-independent properties (batch consistency, graph == eager, run-to-run determinism).

Tolerances (stated per north-star: <= 1e-3 abs on predicted depth; depth RMSE within 1e-3 of the reference):
  fp32 modes (naive_fp32, fp32) and the split-f16 mode f16x3 (f16 operand pairs, three MFMAs per product, fp32 tensors): latent within
  2e-5 * max|x_0| (fp32 round-off class), decoded depth within 1e-3 ABS (max over pixels) on EVERY case -- including the far-range
  golden (depths to 159 m) and the full KITTI size with the decoder shifted to the KITTI depth range (0..80 m).
  bf16 / f16 modes: the decoder ends in exp(-z), so their depth error is RELATIVE (a fixed error in z): measured relative depth RMSE
  4.1e-4 .. 5.8e-4 (bf16 mode) / 1.3e-4 .. 1.9e-4 (f16).  An absolute 1e-3 therefore holds only up to an RMS depth of ~2.4 m (bf16) / ~7 m (f16) with
  these untrained weights, whose loop amplifies the latent to |x_0| ~ 5e2 (a trained denoiser does not; no checkpoint is available
  offline).  Gates: (a) the north star's RMSE reading, depth RMSE <= 1e-3, ASSERTED at the full NYU and KITTI sizes on the near-range
  weight set (depths <= 13 m) plus a max-abs regression bound (bf16 2e-2, f16 5e-3); (b) relative depth RMSE bounds, asserted on every
  case incl. the far range (DEPTH_REL_RMSE_TOL), so that the 16-bit modes cannot drift unnoticed where (a) does not hold.  The depth range
  over which (a) holds is recorded (`abs_rmse_1e3_holds_to_rms_depth_m`) and printed by bench.py.  "bf16" is the library's default bf16
  mode (bf16 MFMA operands on the large convolutions, f16 storage / thin layers); the all-bf16 variant (option
  bf16_storage=1) is measured beside it and only recorded -- it sits at ~1.2e-3, which is why it is not the default.
  f16r (refined f16, DD_PREC_F16R; round 4): the 16-bit mode that is held to the RMSE reading WHERE KITTI LIVES -- depth RMSE <= 1e-3 / 1.5
  (the margin VERDICT r3 asks of a headline mode) at the full NYU and KITTI sizes with the decoder at near range AND shifted to 0..80 m.  f16
  operands with one MFMA per product on the two large convolutions; conv1 / conv3(cond) on split operands, conv4's weights as a stacked f16
  pair, y3 and the hoisted term handed over as block-scaled int16 (y3: one fp32 scale per pixel; the term: one per 32-pixel x 32-cout accumulator block;
  tools/bf16_error_budget.py: those hand-overs and the thin layers, not the large convolutions, made the f16 mode's error).
"""
import numpy as np
import pytest
import torch

from diffusiondepth_amd import synth

pytestmark = pytest.mark.gpu

LATENT_TOL = {"naive_fp32": 2e-5, "fp32": 2e-5, "f16x3": 2e-5, "f16": 1.5e-3, "bf16": 1e-2, "f16r": 8e-4}   # x max|x_0|  (measured: 1e-6, 1e-6, 1e-6, 5e-4, 4e-3; f16r 1.9e-4 emulated)
EPS_TOL = {"naive_fp32": 5e-5, "fp32": 5e-5, "f16x3": 5e-5, "f16": 1.5e-2, "bf16": 1e-1, "f16r": 1e-2}      # abs on eps (values O(1..4); measured 1e-5, 1e-5, 1e-5, 5e-3, 4e-2)
ALL_PREC = ["naive_fp32", "fp32", "f16x3", "bf16", "f16", "f16r"]
ABS_PREC = ("naive_fp32", "fp32", "f16x3")                                      # the modes held to 1e-3 ABS on every pixel of the depth map
DEPTH_RMSE_TOL = 1e-3                                                           # north star, asserted for every precision at full size
DEPTH_MAXABS_TOL = {"fp32": 1e-3, "f16x3": 1e-3, "bf16": 2e-2, "f16": 5e-3, "f16r": 4e-3}     # fp32 / f16x3: the north star's abs reading; 16-bit: regression bounds
DEPTH_REL_RMSE_TOL = {"bf16": 1.2e-3, "f16": 4e-4, "f16r": 2e-4}                # rms of (d - d_ref) / d_ref, every case incl. far range: 2x the worst measured (5.8e-4 / 1.9e-4 / 9.9e-5 on loop_res_far)
HEADLINE_MARGIN = 1.5                                                           # f16r: depth RMSE <= 1e-3 / 1.5 near AND at KITTI's depth range


def rel_rmse(d, dref):
    r = (np.asarray(d, np.float64) - np.asarray(dref, np.float64)) / np.maximum(np.asarray(dref, np.float64), 1e-6)
    return float(np.sqrt((r * r).mean()))


def rms_depth(dref):
    return float(np.sqrt((np.asarray(dref, np.float64) ** 2).mean()))


@pytest.fixture(scope="module")
def U():
    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a HIP device: the product has no CPU fallback")
    import gpu_util
    return gpu_util


def test_native_library_is_loaded_and_reports_gfx950(U):
    import diffusiondepth_amd as dda
    be = dda.HipDenoiser()
    assert "gfx950" in be.version
    with pytest.raises(RuntimeError, match="not committed"):
        be.denoise(torch.zeros(1, 16, 8, 8, device="cuda"), torch.zeros(1, 256, 8, 8, device="cuda"), 5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        be.denoise(torch.zeros(1, 16, 8, 8), torch.zeros(1, 256, 8, 8), 5)
    be.close()


def test_add_noise_matches_reference(U, golden, cases):
    c, g = cases["sched"], golden("sched")
    rs = np.random.RandomState(c["seed"])
    rs.standard_normal(c["shape"]); rs.standard_normal(c["shape"])
    B = len(c["add_noise_t"])
    x0 = rs.standard_normal((B,) + tuple(c["shape"][1:])).astype(np.float32)
    nz = rs.standard_normal((B,) + tuple(c["shape"][1:])).astype(np.float32)
    be = U.backend_for(cases["loop_res"])
    out = be.add_noise(U.cu(x0), U.cu(nz), torch.tensor(c["add_noise_t"], device="cuda")).cpu().numpy()
    assert U.maxabs(out, g["add_noise"]) < 1e-6


def test_codec_matches_reference(U, golden, cases):
    c, g = cases["codec"], golden("codec")
    be = U.backend_for(c)
    for i, (B, H, W) in enumerate(c["sizes"]):
        gt = synth.make_gt_depth(c["iseed"] + i, B, H, W)
        lat = be.encode(U.cu(gt)).cpu().numpy()
        assert lat.shape == g[f"latent_{i}"].shape
        e = U.maxabs(lat, g[f"latent_{i}"])
        h, w = synth.latent_hw(H, W)
        z = np.random.RandomState(c["iseed"] + 100 + i).standard_normal((B, 16, h, w)).astype(np.float32) * c["latent_scale"]
        d = be.decode(U.cu(z)).cpu().numpy()
        dref = g[f"depth_{i}"]
        rel = float((np.abs(d - dref) / np.maximum(np.abs(dref), 1e-2)).max())
        U.record("codec", case=i, enc_maxabs=e, dec_maxrel=rel)
        assert e < 2e-5
        assert d.shape == dref.shape and rel < 5e-5


@pytest.mark.parametrize("prec", ALL_PREC)
def test_single_denoiser_call_vs_reference(U, golden, cases, prec):
    from oracle import ddim_oracle as O
    c, g = cases["denoise_res"], golden("denoise_res")
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
    eps_b = be.denoise_once(x, U.cu(inp["timesteps"]), cond, prec)
    # per-layer raw conv outputs against the oracle, to localise a failing layer
    sd = U.sd_for(c)
    sd64 = {k: v.astype(np.float64) for k, v in sd.items()}
    y1_ref = O.conv2d(inp["x_T"].astype(np.float64), sd64["model.noise_embedding.0.weight"], sd64["model.noise_embedding.0.bias"])
    layer_err = {"y1": U.maxabs(be.debug_fetch("y1", c["B"], c["h"], c["w"]).cpu().numpy(), y1_ref) / np.abs(y1_ref).max()}
    eps_b = eps_b.cpu().numpy()
    eps_s = be.denoise_once(x, torch.tensor(c["t"], device="cuda"), cond, prec).cpu().numpy()
    eb, es = U.maxabs(eps_b, g["eps_batch_t"]), U.maxabs(eps_s, g["eps_scalar_t"])
    U.record("denoise_once", prec=prec, eps_batch_maxabs=eb, eps_scalar_maxabs=es, eps_rms=U.rms(eps_b, g["eps_batch_t"]), **layer_err)
    assert eps_b.min() >= 0.0
    assert layer_err["y1"] < (1e-5 if prec in ABS_PREC else 2e-2)
    assert eb < EPS_TOL[prec] and es < EPS_TOL[prec], (eb, es)


@pytest.mark.parametrize("prec", ALL_PREC)
@pytest.mark.parametrize("name", ["loop_res", "loop_res_far"])
def test_ddim_loop_vs_reference_golden(U, golden, cases, name, prec):
    c, g = cases[name], golden(name)
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    for T in c["T"]:
        x0 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), T, prec)
        depth = be.decode(x0).cpu().numpy()
        x0 = x0.cpu().numpy()
        ref, dref = g[f"x0_T{T}"], g[f"depth_T{T}"]
        scale = float(np.abs(ref).max())
        e = U.maxabs(x0, ref)
        de = U.maxabs(depth, dref)
        drel = float((np.abs(depth - dref) / np.maximum(dref, 1e-2)).max())
        rr = rel_rmse(depth, dref)
        U.record("loop", case=name, prec=prec, T=T, latent_maxabs=e, latent_scale=scale, latent_rms=U.rms(x0, ref),
                 depth_maxabs=de, depth_rmse=U.rms(depth, dref), depth_maxrel=drel, depth_max=float(dref.max()),
                 depth_rel_rmse=rr, depth_rms=rms_depth(dref))
        assert np.isfinite(x0).all()
        assert e < LATENT_TOL[prec] * scale, (e, scale)
        if prec in ABS_PREC:
            assert de < 1e-3, de                       # the north-star tolerance on predicted depth, near AND far range (159 m)
        else:
            assert rr < DEPTH_REL_RMSE_TOL[prec], (name, prec, rr)      # 16-bit modes: the error is relative; absolute 1e-3 only at short range


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "bf16", "f16r"])
def test_ragged_sizes_and_batch_vs_oracle(U, prec):
    """Tile edges: sizes that are not multiples of the 8x32 tile, 1-pixel-wide, taller than wide, B=3."""
    from oracle import ddim_oracle as O
    c = {"wseed": 7244}
    be = U.backend_for(c)
    sd = U.sd_for(c)
    for (B, h, w, T) in [(3, 9, 33, 3), (1, 1, 1, 2), (2, 17, 5, 3), (1, 8, 32, 2), (1, 40, 70, 2),
                         (1, 8, 19, 7), (1, 8, 19, 1)]:      # step counts that do not divide 1000 (ratio 142), a single step
        inp = synth.make_inputs(100 + h, B, h, w)
        x0 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), T, prec).cpu().numpy()
        ref = O.ddim_loop(sd, inp["x_T"], inp["cond"], T)
        scale = float(np.abs(ref).max())
        e = U.maxabs(x0, ref)
        U.record("ragged", prec=prec, B=B, h=h, w=w, T=T, latent_maxabs=e, latent_scale=scale)
        assert e < LATENT_TOL[prec] * scale, (B, h, w, e, scale)


def test_graph_equals_eager_and_is_deterministic(U, cases):
    c = cases["loop_res"]
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], 2, 24, 40)
    x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
    for prec, tol in (("fp32", 1e-5), ("bf16", 1e-5)):
        be.set_option("graph", 1)
        g0 = be.counter("graph_launches")
        a = be.denoise(x, cond, 20, prec).cpu().numpy()
        b = be.denoise(x, cond, 20, prec).cpu().numpy()
        lanes = min(getattr(be, "n_streams", 1), x.shape[0])         # one graph per concurrent lane (DDEPTH_STREAMS, default 2)
        assert be.counter("graph_launches") - g0 == 2 * lanes and be.counter("graph_capture_failures") == 0
        be.set_option("graph", 0)
        e0 = be.counter("eager_loops")
        cc = be.denoise(x, cond, 20, prec).cpu().numpy()
        assert be.counter("eager_loops") - e0 == lanes
        be.set_option("graph", 1)
        scale = np.abs(a).max()
        # fp64 atomics make the GroupNorm sums order-dependent only at the 1e-16 level
        U.record("graph_vs_eager", prec=prec, rerun_maxabs=U.maxabs(a, b), eager_maxabs=U.maxabs(a, cc), scale=float(scale))
        assert U.maxabs(a, b) <= tol * scale and U.maxabs(a, cc) <= tol * scale


def test_batch_consistency(U, cases):
    """Independent samples: a batch of identical inputs gives identical per-sample outputs, and a
    sample's result does not depend on its batch neighbours (GroupNorm is per-sample)."""
    c = cases["loop_res"]
    be = U.backend_for(c)
    one = synth.make_inputs(5, 1, 19, 37)
    other = synth.make_inputs(6, 1, 19, 37)
    x = U.cu(np.concatenate([one["x_T"], other["x_T"], one["x_T"]]))
    cond = U.cu(np.concatenate([one["cond"], other["cond"], one["cond"]]))
    for prec in ("fp32", "bf16"):
        out = be.denoise(x, cond, 4, prec).cpu().numpy()
        solo = be.denoise(U.cu(one["x_T"]), U.cu(one["cond"]), 4, prec).cpu().numpy()
        s = np.abs(solo).max()
        assert U.maxabs(out[0], out[2]) <= 1e-6 * s
        assert U.maxabs(out[0], solo[0]) <= 1e-6 * s


def test_head_forward_matches_reference_golden(U, golden, cases):
    """Whole DDIMDepthEstimate_Res.forward (PyTorch FPN + HIP encoder/loop/decoder/ddim_loss) with
    the reference's RNG draws injected, against the output of the reference head class."""
    import diffusiondepth_amd as dda
    c, g = cases["head_res"], golden("head_res")
    sd = synth.make_state_dict(c["wseed"], "res", c["decoder_gain"], c["decoder_log_scale"])
    sd.update(synth.make_fpn_state_dict(c["fseed"]))
    head = dda.DDIMDepthEstimate_Res(in_channels=[64, 128, 256, 512], inference_steps=c["T"], num_train_timesteps=1000,
                                     depth_feature_dim=16, loss_cfgs=[], precision="fp32").eval()
    missing, unexpected = head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    head = head.cuda()
    B, H, W = c["B"], c["H"], c["W"]
    fp = [U.cu(f) for f in synth.make_backbone_features(c["iseed"], B, H, W)]
    gt = U.cu(synth.make_gt_depth(c["iseed"] + 1, B, H, W))
    h, w = synth.latent_hw(H, W)
    inp = synth.make_inputs(c["iseed"] + 2, B, h, w)
    draws = [U.cu(inp["x_T"]), torch.from_numpy(inp["noise"])]
    real_randn, real_randint = torch.randn, torch.randint
    torch.randn = lambda *a, **k: draws.pop(0)
    torch.randint = lambda *a, **k: U.cu(inp["timesteps"])
    try:
        with torch.no_grad():
            out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
    finally:
        torch.randn, torch.randint = real_randn, real_randint
    assert set(out) == set(c["output_keys"])
    pred = out["pred"].cpu().numpy()
    e_pred, e_init = U.maxabs(pred, g["pred"]), U.maxabs(out["pred_init"].cpu().numpy(), g["pred_init"])
    e_loss = abs(float(out["ddim_loss"]) - float(g["ddim_loss"][0]))
    U.record("head_res", pred_maxabs=e_pred, pred_init_maxabs=e_init, ddim_loss_abs=e_loss, pred_max=float(g["pred"].max()))
    assert e_init < 2e-5
    assert e_pred < 1e-3                      # north-star: <= 1e-3 abs on predicted depth
    assert e_loss < 1e-4 * max(1.0, abs(float(g["ddim_loss"][0])))


@pytest.mark.parametrize("size", ["nyu", "kitti"])
def test_full_size_loop_vs_torch_cpu_port(U, size):
    """BASELINE.json configs 2/3 at full size: HIP fp32 / bf16 / f16 loops against the torch-CPU port; the depth gates of the
    north star are asserted for every precision that bench.py can time."""
    import time
    from oracle import torch_cpu_port as P
    h, w = {"nyu": (114, 152), "kitti": (176, 608)}[size]
    c = {"wseed": 7240}
    be = U.backend_for(c)
    sd = U.sd_for(c)
    inp = synth.make_inputs(77, 1, h, w)
    t0 = time.time()
    ref = P.ddim_loop(P.to_torch_sd(sd), inp["x_T"], inp["cond"], 20)
    cpu_s = time.time() - t0
    dref = P.decode(P.to_torch_sd(sd), ref).numpy()
    ref = ref.numpy()
    scale = float(np.abs(ref).max())
    x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
    # the same loop result decoded at KITTI's depth range: the decoder's last bias shifted by -FAR_LOG_SCALE multiplies every depth by
    # e^1.8 (~0.5 .. 80 m); only the decoder differs, so the latent reference is shared
    FAR_LOG_SCALE = 1.8
    cfar = {"wseed": 7240, "decoder_log_scale": FAR_LOG_SCALE}
    be_far, sd_far = U.backend_for(cfar), U.sd_for(cfar)
    dref_far = P.decode(P.to_torch_sd(sd_far), torch.from_numpy(ref)).numpy()
    for prec in ("fp32", "f16x3", "bf16", "f16", "f16r"):
        x0 = be.denoise(x, cond, 20, prec)
        d = be.decode(x0).cpu().numpy()
        d_far = be_far.decode(x0).cpu().numpy()
        x0 = x0.cpu().numpy()
        e, de = U.maxabs(x0, ref), U.maxabs(d, dref)
        rr = rel_rmse(d, dref)
        U.record("full_size", size=size, prec=prec, latent_maxabs=e, latent_scale=scale, latent_rms=U.rms(x0, ref),
                 depth_maxabs=de, depth_rmse=U.rms(d, dref), depth_max=float(dref.max()), cpu_port_seconds=cpu_s,
                 depth_rel_rmse=rr, depth_rms=rms_depth(dref), abs_rmse_1e3_holds_to_rms_depth_m=1e-3 / max(rr, 1e-12),
                 far_depth_max=float(dref_far.max()), far_depth_rms=rms_depth(dref_far), far_depth_rmse=U.rms(d_far, dref_far),
                 far_depth_maxabs=U.maxabs(d_far, dref_far), far_depth_rel_rmse=rel_rmse(d_far, dref_far))
        assert e < LATENT_TOL[prec] * scale * (2.5 if prec in ABS_PREC else 1.0), (prec, e, scale)   # port itself is fp32
        assert U.rms(d, dref) <= DEPTH_RMSE_TOL, (size, prec, U.rms(d, dref))
        assert de <= DEPTH_MAXABS_TOL[prec], (size, prec, de)
        if prec in ABS_PREC:
            # the abs-clean modes keep 1e-3 on every pixel at KITTI's depth range too
            assert dref_far.max() > 40.0 and U.maxabs(d_far, dref_far) <= 1e-3, (size, prec, U.maxabs(d_far, dref_far))   # NYU 52 m, KITTI 69 m
        else:
            assert rel_rmse(d_far, dref_far) < DEPTH_REL_RMSE_TOL[prec], (size, prec, rel_rmse(d_far, dref_far))
        if prec == "f16r":
            # the headline mode: the north star's RMSE reading with margin, on this workload and with the same latents decoded at 0..80 m
            assert dref_far.max() > 40.0
            assert U.rms(d, dref) <= DEPTH_RMSE_TOL / HEADLINE_MARGIN, (size, prec, U.rms(d, dref))
            assert U.rms(d_far, dref_far) <= DEPTH_RMSE_TOL / HEADLINE_MARGIN, (size, prec, U.rms(d_far, dref_far))
    # the all-bf16 variant beside the default bf16 mode: recorded, not gated (it is the reason the default stores f16)
    import diffusiondepth_amd as dda
    pure = dda.HipDenoiser()
    pure.load_state_dict(sd)
    pure.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    pure.set_option("bf16_storage", 1)
    d = pure.decode(pure.denoise(x, cond, 20, "bf16")).cpu().numpy()
    U.record("full_size", size=size, prec="bf16_all_bf16_tensors", depth_maxabs=U.maxabs(d, dref), depth_rmse=U.rms(d, dref), depth_max=float(dref.max()))
    assert U.rms(d, dref) <= 3e-3                     # sanity only
    pure.close()


def test_big_tile_and_lane_paths_agree_with_the_single_image_path_at_kitti_size(U):
    """At KITTI size a batch takes other kernels than one image: two concurrent lanes (option streams), the hoisted conv3 in its one-patch-buffer
    form (836 tiles per two-image lane against 512 resident slots; on request -- option big_tiles -- on 16x32 tiles, kernel ids 48 / 49), conv4's
    streaming workgroups walking several tiles each.  Every image of a batch of 3 must equal its solo run (8x32 tiles, one tile per streaming workgroup) within the precision's
    bound (bf16: the 16-bit rounding class -- other GroupNorm partial-sum order; fp32 and f16x3 do not have the big tiles: round-off), and
    the forced settings must agree with the automatic ones."""
    c = {"wseed": 7240}
    be = U.backend_for(c)
    h, w, T = 176, 608, 4
    inp = synth.make_inputs(91, 3, h, w)
    x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
    for prec in ("bf16", "fp32", "f16r"):      # f16r: the hoisted term reformatted into the 16x32-tile order (block-scaled int16) and the stacked conv4 reading int16 y3 + per-pixel scales
        batch = be.denoise(x, cond, T, prec).cpu().numpy()
        scale = float(np.abs(batch).max())
        for i in range(3):
            solo = be.denoise(x[i:i + 1].contiguous(), cond[i:i + 1].contiguous(), T, prec).cpu().numpy()
            e = U.maxabs(batch[i:i + 1], solo)
            U.record("batch_vs_solo_kitti", prec=prec, image=i, maxabs=e, scale=scale)
            assert e < (2e-6 if prec == "fp32" else LATENT_TOL[prec]) * scale, (prec, i, e, scale)
    try:
        be.set_option("big_tiles", 0)
        small = be.denoise(x, cond, T, "bf16").cpu().numpy()
        be.set_option("big_tiles", 1)
        big = be.denoise(x, cond, T, "bf16").cpu().numpy()
    finally:
        be.set_option("big_tiles", -1)
    auto = be.denoise(x, cond, T, "bf16").cpu().numpy()
    s = float(np.abs(auto).max())
    assert U.maxabs(small, big) < LATENT_TOL["bf16"] * s
    # the automatic rule (round 6): the Res denoiser keeps its 8x32 tiles -- one-patch-buffer form when a plan's tiles exceed the resident slots -- whatever
    # the lane count (the 16x32 form under lanes lost 4-6 % on every box: profiles/r06_experiments.md section 2); per image the two 8x32 forms are bit-identical
    assert np.array_equal(auto, small)


def test_refined_f16_reads_an_explicit_condition_tensor_in_place(U):
    """Option "cond_direct" (default on): in the refined mode the once-per-image conv3(cond) reads the caller's NCHW fp32 tensor itself (kernel id
    CONV3C_NCHW: eight 4-byte loads per staging item) instead of a channel-blocked copy made first.  Same values, same arithmetic: bit-identical
    x_0 at KITTI size (two lanes, 16x32-tile consumers) and on a ragged size, loop and single call."""
    be = U.backend_for({"wseed": 7240})
    for B, h, w, T in ((3, 176, 608, 3), (2, 45, 75, 2)):
        inp = synth.make_inputs(92, B, h, w)
        x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
        t = torch.full((B,), 321, device="cuda", dtype=torch.long)
        try:
            be.set_option("cond_direct", 0)
            a, ea = be.denoise(x, cond, T, "f16r").cpu().numpy(), be.denoise_once(x, t, cond, "f16r").cpu().numpy()
        finally:
            be.set_option("cond_direct", 1)
        b, eb = be.denoise(x, cond, T, "f16r").cpu().numpy(), be.denoise_once(x, t, cond, "f16r").cpu().numpy()
        assert np.isfinite(b).all() and np.array_equal(a, b) and np.array_equal(ea, eb), (B, h, w)


@pytest.mark.parametrize("variant", ["res", "swin"])
def test_a_nan_in_the_inputs_poisons_that_image_and_only_that_image(U, variant):
    """The reference turns ONE NaN / Inf of an image's condition map or x_T into an all-NaN prediction for that image (GroupNorm spreads it, torch.relu keeps
    it).  The kernels' ReLU (v_max_f32) drops a NaN operand, so the library carries it through the GroupNorm statistics instead:
    every precision, the loop at KITTI size (two lanes, 16x32 tiles) and on a ragged size, single calls, and the decoder; the clean image of the batch is
    bit-identical to its clean run."""
    c = {"wseed": 7240, "variant": variant}
    be = U.backend_for(c)
    for (B, h, w, T) in ((2, 176, 608, 3), (2, 9, 33, 2)):
        chw = None if variant == "res" else ((h + 1) // 2, (w + 1) // 2)
        inp = synth.make_inputs(321 + h, B, h, w, chw)
        x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
        bad_c = cond.clone()
        bad_c[1, 7, bad_c.shape[2] // 2, 5] = float("nan")
        bad_x = x.clone()
        bad_x[0, 3, h // 2, w // 3] = float("inf")
        for prec in ("f16r", "f16", "bf16", "f16x3", "fp32"):
            clean = be.denoise(x, cond, T, prec)
            assert torch.isfinite(clean).all(), prec
            out = be.denoise(x, bad_c, T, prec)
            assert torch.isnan(out[1]).all(), (variant, prec, h, "a NaN of the condition map was lost", int(torch.isnan(out[1]).sum()))
            assert torch.equal(out[0], clean[0]), (variant, prec, h, "the clean image of the batch changed")
            out = be.denoise(bad_x, cond, T, prec)
            assert torch.isnan(out[0]).all() and torch.equal(out[1], clean[1]), (variant, prec, h, "an Inf of x_T was lost")
            eps = be.denoise_once(x, torch.full((B,), 321, device="cuda", dtype=torch.long), bad_c, prec)
            assert torch.isnan(eps[1]).all() and torch.isfinite(eps[0]).all(), (variant, prec, h, "single call")
        d = be.decode(out)
        assert torch.isnan(d[0]).all() and torch.isfinite(d[1]).all()


# ---- Swin / MPViT variant of the denoiser: UpSample_add fuse, stride-4 condition map ----
@pytest.mark.parametrize("prec", ["fp32", "f16x3", "bf16", "f16"])
def test_swin_variant_single_call_and_loop_vs_reference(U, golden, cases, prec):
    c, g = cases["denoise_swin"], golden("denoise_swin")
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], c["cond_hw"])
    x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
    eps_b = be.denoise_once(x, U.cu(inp["timesteps"]), cond, prec).cpu().numpy()
    eps_s = be.denoise_once(x, torch.tensor(c["t"], device="cuda"), cond, prec).cpu().numpy()
    eb, es = U.maxabs(eps_b, g["eps_batch_t"]), U.maxabs(eps_s, g["eps_scalar_t"])
    c2, g2 = cases["loop_swin"], golden("loop_swin")
    assert U.backend_for(c2) is be
    inp2 = synth.make_inputs(c2["iseed"], c2["B"], c2["h"], c2["w"], c2["cond_hw"])
    x0 = be.denoise(U.cu(inp2["x_T"]), U.cu(inp2["cond"]), 20, prec)
    depth = be.decode(x0).cpu().numpy()
    x0 = x0.cpu().numpy()
    ref, dref = g2["x0_T20"], g2["depth_T20"]
    scale = float(np.abs(ref).max())
    e, de = U.maxabs(x0, ref), U.maxabs(depth, dref)
    U.record("swin", prec=prec, eps_batch_maxabs=eb, eps_scalar_maxabs=es, latent_maxabs=e, latent_scale=scale,
             depth_maxabs=de, depth_rmse=U.rms(depth, dref), depth_max=float(dref.max()))
    assert eps_b.min() >= 0.0
    assert eb < EPS_TOL[prec] and es < EPS_TOL[prec], (eb, es)
    assert e < LATENT_TOL[prec] * scale, (e, scale)
    if prec in ABS_PREC:
        assert de < 1e-3
    # depth gates of the Swin denoiser: f16 is its inference mode inside the tolerance (BASELINE config 5 names fp16); bf16, with two more
    # 256 -> 256 convolutions on bf16 operands per step, is NOT -- 1.4e-3 here, 1.6e-3 at KITTI size with the step-invariant terms hoisted
    # out of those operands (round 3; 2.6e-3 / 3.4e-3 before: DESIGN.md section 4) -- it is the training precision, bounded at twice the
    # measured value so that it cannot drift unnoticed
    if prec == "f16":
        assert U.rms(depth, dref) <= DEPTH_RMSE_TOL
    if prec == "bf16":
        assert U.rms(depth, dref) <= 3e-3


def test_swin_refined_f16_mode_at_kitti_depth_range(U, golden, cases):
    """DD_PREC_F16R on the Swin / MPViT denoiser (hoisted forward-only plans): conv1 / conv4 / the once-per-image chain on f16 weight pairs, y3 and
    the hoisted term as block-scaled int16, the two large convolutions (convA', the 5x5 form) on plain f16 operands.  Against the reference's golden
    loop, ragged sizes against the oracle, and -- the point of the mode -- at the full KITTI size with the decoder shifted to 0..80 m, where the f16
    mode is OUTSIDE the 1e-3 depth RMSE (tools/swin_error_budget.py: 1.16e-3 emulated) and this one inside (6.2e-4 emulated)."""
    from oracle import ddim_oracle as O
    from oracle import torch_cpu_port as P
    c2, g2 = cases["loop_swin"], golden("loop_swin")
    be, sd = U.backend_for(c2), U.sd_for(c2)
    inp = synth.make_inputs(c2["iseed"], c2["B"], c2["h"], c2["w"], c2["cond_hw"])
    ref, dref = g2["x0_T20"], g2["depth_T20"]
    scale = float(np.abs(ref).max())
    x0 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), 20, "f16r")
    d = be.decode(x0).cpu().numpy()
    x16 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), 20, "f16")
    d16 = be.decode(x16).cpu().numpy()
    e = U.maxabs(x0.cpu().numpy(), ref)
    U.record("swin_f16r", case="loop_swin", latent_maxabs=e, latent_scale=scale, depth_rmse=U.rms(d, dref), depth_rmse_f16=U.rms(d16, dref), depth_max=float(dref.max()))
    assert e < LATENT_TOL["f16r"] * scale and U.rms(d, dref) < 0.75 * U.rms(d16, dref), (e, scale, U.rms(d, dref), U.rms(d16, dref))
    for (B, h, w, ch, cw, T) in [(2, 5, 40, 3, 9, 3), (1, 13, 6, 5, 3, 2), (2, 11, 19, 6, 10, 2)]:
        i = synth.make_inputs(400 + h, B, h, w, (ch, cw))
        rr = O.ddim_loop(sd, i["x_T"], i["cond"], T, "swin")
        xr = be.denoise(U.cu(i["x_T"]), U.cu(i["cond"]), T, "f16r").cpu().numpy()
        assert U.maxabs(xr, rr) < LATENT_TOL["f16r"] * float(np.abs(rr).max()), (B, h, w)
    # ONE refined-f16 call with PER-SAMPLE timesteps (round 6: dd_denoise_once runs the hoisted form with one E[t] border table per image): against the
    # fp64 oracle on ragged shapes whose border classes are all active, timesteps that differ per image, and closer to it than the plain f16 kernels
    for (B, h, w, ch, cw) in [(3, 9, 33, 5, 17), (2, 24, 40, 12, 20), (1, 3, 5, 2, 3)]:
        i = synth.make_inputs(410 + h, B, h, w, (ch, cw))
        tt = np.array([37, 950, 512][:B], dtype=np.int64)
        ref_eps = O.denoiser_forward(sd, i["x_T"], tt, i["cond"], "swin")
        er = be.denoise_once(U.cu(i["x_T"]), torch.from_numpy(tt).cuda(), U.cu(i["cond"]), "f16r").cpu().numpy()
        e16 = be.denoise_once(U.cu(i["x_T"]), torch.from_numpy(tt).cuda(), U.cu(i["cond"]), "f16").cpu().numpy()
        U.record("swin_f16r_single_call", B=B, h=h, w=w, eps_maxabs=U.maxabs(er, ref_eps), eps_maxabs_f16=U.maxabs(e16, ref_eps))
        assert U.maxabs(er, ref_eps) < EPS_TOL["f16r"], (B, h, w, U.maxabs(er, ref_eps))
    # full KITTI size, the decoder shifted so that the depths span KITTI's 0..80 m
    h, w = 176, 608
    SWIN_LOG_SCALE = 1.25            # (this denoiser's near-range weights decode up to ~23 m; x e^1.25 -> ~80 m)
    c = {"wseed": 7240, "variant": "swin"}
    cfar = {"wseed": 7240, "variant": "swin", "decoder_log_scale": SWIN_LOG_SCALE}
    bek, sdk = U.backend_for(c), U.sd_for(c)
    be_far, sd_far = U.backend_for(cfar), U.sd_for(cfar)
    ik = synth.make_inputs(78, 1, h, w, (88, 304))
    lat = P.ddim_loop(P.to_torch_sd(sdk), ik["x_T"], ik["cond"], 20, variant="swin")
    dk, dk_far = P.decode(P.to_torch_sd(sdk), lat).numpy(), P.decode(P.to_torch_sd(sd_far), lat).numpy()
    out = {}
    for prec in ("f16", "f16r"):
        xk = bek.denoise(U.cu(ik["x_T"]), U.cu(ik["cond"]), 20, prec)
        out[prec] = (U.rms(bek.decode(xk).cpu().numpy(), dk), U.rms(be_far.decode(xk).cpu().numpy(), dk_far))
        U.record("swin_full_size", prec=prec, depth_rmse=out[prec][0], far_depth_rmse=out[prec][1], depth_max=float(dk.max()), far_depth_max=float(dk_far.max()))
    assert 60.0 < dk_far.max() < 110.0
    assert out["f16r"][0] <= DEPTH_RMSE_TOL / HEADLINE_MARGIN and out["f16r"][1] <= DEPTH_RMSE_TOL, out       # inside the tolerance at KITTI's range ...
    assert out["f16r"][1] < 0.75 * out["f16"][1], out                                                               # ... where the f16 mode is not (recorded)


def test_swin_loop_at_the_step_count_of_baseline_config_5(U):
    """BASELINE config 5 / `--inference_steps 50` (src/config.py:136-139; scheduling_ddim.py:215-229: timesteps 980, 960, ..., 0): the 200-node graph, the
    c1c2 table and the hoisted plans' `ttab[T]` at T = 50 -- (a) every precision against the fp64 oracle on ragged sizes (the hoisted 5x5 form's border
    classes all active), (b) the shipped mode f16r, f16 and the abs-clean f16x3 at the full KITTI size against the torch-CPU port, near range and with the
    decoder shifted to 0..80 m.  VERDICT r4 missing #2."""
    import time
    from oracle import ddim_oracle as O
    from oracle import torch_cpu_port as P
    T = 50
    import diffusiondepth_amd as dda
    sch = dda.DDIMScheduler()
    sch.set_timesteps(T)
    assert [int(t) for t in sch.timesteps[:3]] == [980, 960, 940] and int(sch.timesteps[-1]) == 0 and len(sch.timesteps) == T
    c = {"wseed": 7240, "variant": "swin"}
    be, sd = U.backend_for(c), U.sd_for(c)
    for (B, h, w, ch, cw) in [(2, 9, 21, 5, 11), (1, 14, 7, 7, 4)]:
        i = synth.make_inputs(500 + h, B, h, w, (ch, cw))
        rr = O.ddim_loop(sd, i["x_T"], i["cond"], T, "swin")
        sc = float(np.abs(rr).max())
        for prec in ("fp32", "f16x3", "f16", "f16r", "bf16"):
            xr = be.denoise(U.cu(i["x_T"]), U.cu(i["cond"]), T, prec).cpu().numpy()
            e = U.maxabs(xr, rr)
            U.record("swin_T50_ragged", prec=prec, B=B, h=h, w=w, latent_maxabs=e, latent_scale=sc)
            assert np.isfinite(xr).all() and e < LATENT_TOL[prec] * sc * (2.5 if prec in ABS_PREC else 2.0), (prec, B, h, w, e, sc)
    # Res denoiser at T = 50 too (ragged, the oracle): the c1c2 table of 50 entries through conv1's prologue
    cr = {"wseed": 7240}
    ber, sdr = U.backend_for(cr), U.sd_for(cr)
    i = synth.make_inputs(511, 2, 9, 21)
    rr = O.ddim_loop(sdr, i["x_T"], i["cond"], T)
    for prec in ("fp32", "f16r"):
        e = U.maxabs(ber.denoise(U.cu(i["x_T"]), U.cu(i["cond"]), T, prec).cpu().numpy(), rr)
        assert e < LATENT_TOL[prec] * float(np.abs(rr).max()) * 2.5, (prec, e)
    # full KITTI size
    h, w = 176, 608
    SWIN_LOG_SCALE = 1.25
    cfar = {"wseed": 7240, "variant": "swin", "decoder_log_scale": SWIN_LOG_SCALE}
    be_far, sd_far = U.backend_for(cfar), U.sd_for(cfar)
    ik = synth.make_inputs(79, 1, h, w, (88, 304))
    t0 = time.time()
    lat = P.ddim_loop(P.to_torch_sd(sd), ik["x_T"], ik["cond"], T, variant="swin")
    cpu_s = time.time() - t0
    dk, dk_far = P.decode(P.to_torch_sd(sd), lat).numpy(), P.decode(P.to_torch_sd(sd_far), lat).numpy()
    n0 = be.counter("graph_launches")
    res = {}
    for prec in ("f16x3", "f16", "f16r"):
        xk = be.denoise(U.cu(ik["x_T"]), U.cu(ik["cond"]), T, prec)
        d, dfar = be.decode(xk).cpu().numpy(), be_far.decode(xk).cpu().numpy()
        res[prec] = dict(depth_rmse=U.rms(d, dk), depth_maxabs=U.maxabs(d, dk), far_depth_rmse=U.rms(dfar, dk_far), far_depth_maxabs=U.maxabs(dfar, dk_far))
        U.record("swin_T50_kitti", prec=prec, depth_max=float(dk.max()), far_depth_max=float(dk_far.max()), cpu_port_seconds=cpu_s, **res[prec])
    assert be.counter("graph_launches") >= n0 + 3, "the T = 50 loop did not run as a hipGraph"
    assert res["f16x3"]["depth_maxabs"] <= 1e-3 and res["f16x3"]["far_depth_maxabs"] <= 1e-3, res           # abs-clean at 50 steps too
    assert res["f16r"]["depth_rmse"] <= DEPTH_RMSE_TOL / HEADLINE_MARGIN and res["f16r"]["far_depth_rmse"] <= DEPTH_RMSE_TOL, res
    assert res["f16r"]["far_depth_rmse"] < res["f16"]["far_depth_rmse"], res


def test_swin_variant_odd_sizes_vs_oracle(U):
    """Condition map at a non-integer scale of the latent (as Swin stride-4 maps are: 57x76 -> 114x152)."""
    from oracle import ddim_oracle as O
    c = {"wseed": 7245, "variant": "swin"}
    be = U.backend_for(c)
    sd = U.sd_for(c)
    for (B, h, w, ch, cw, T) in [(2, 11, 19, 6, 10, 2), (1, 20, 36, 10, 18, 2)]:
        inp = synth.make_inputs(300 + h, B, h, w, (ch, cw))
        x0 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), T, "fp32").cpu().numpy()
        ref = O.ddim_loop(sd, inp["x_T"], inp["cond"], T, "swin")
        scale = float(np.abs(ref).max())
        e = U.maxabs(x0, ref)
        U.record("swin_ragged", B=B, h=h, w=w, ch=ch, cw=cw, latent_maxabs=e, latent_scale=scale)
        assert e < LATENT_TOL["fp32"] * scale, (B, h, w, e, scale)


def test_swin_with_and_without_hoisting_the_step_invariant_terms(U, golden, cases):
    """Swin forward-only plans take the condition map and the time embedding through pred.0(convB(convA(.))) outside the loop (kernel ids
    SWIN_CONVA_H / SWIN_PRED_H; default in the 2-byte modes, option hoist_cond = 1 also in fp32).  Both forms against the reference's golden
    loop and, on ragged sizes whose borders exercise every class of the E[t] table (an axis shorter than seven pixels included), the oracle."""
    from oracle import ddim_oracle as O
    c2, g2 = cases["loop_swin"], golden("loop_swin")
    be = U.backend_for(c2)
    sd = U.sd_for(c2)
    inp = synth.make_inputs(c2["iseed"], c2["B"], c2["h"], c2["w"], c2["cond_hw"])
    ref, dref = g2["x0_T20"], g2["depth_T20"]
    scale = float(np.abs(ref).max())
    rag = [(2, 5, 40, 3, 9, 3), (1, 13, 6, 5, 3, 2)]
    rag_in = [synth.make_inputs(400 + h, B, h, w, (ch, cw)) for (B, h, w, ch, cw, T) in rag]
    rag_ref = [O.ddim_loop(sd, i["x_T"], i["cond"], r[5], "swin") for i, r in zip(rag_in, rag)]
    try:
        # (hoist, w5): the reference's order; hoisted with convB and pred.0 as two kernels; hoisted with pred.0 o convB as one 5x5 convolution (default)
        for hoist, w5 in ((0, 1), (1, 0), (1, 1)):
            be.set_option("hoist_cond", hoist)
            be.set_option("swin_w5", w5)
            for prec in ("fp32", "bf16", "f16", "f16x3"):
                if prec == "f16x3" and hoist == 1 and w5 == 0:
                    continue      # (the split mode has no two-kernel hoisted form: it would run the reference's order again)
                x0 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), 20, prec)
                depth = be.decode(x0).cpu().numpy()
                e = U.maxabs(x0.cpu().numpy(), ref)
                er = []
                for i, r, rr in zip(rag_in, rag, rag_ref):
                    xr = be.denoise(U.cu(i["x_T"]), U.cu(i["cond"]), r[5], prec).cpu().numpy()
                    er.append(U.maxabs(xr, rr) / float(np.abs(rr).max()))
                U.record("swin_hoist_ab", hoist=hoist, w5=w5, prec=prec, latent_maxabs=e, latent_scale=scale, depth_rmse=U.rms(depth, dref),
                         depth_maxabs=U.maxabs(depth, dref), ragged_rel=er)
                assert e < LATENT_TOL[prec] * scale and max(er) < LATENT_TOL[prec], (hoist, w5, prec, e, scale, er)
                if prec in ABS_PREC:
                    assert U.maxabs(depth, dref) < 1e-3
    finally:
        be.set_option("hoist_cond", -1)
        be.set_option("swin_w5", 1)


def test_swin_hoisted_tables_follow_a_parameter_update(U):
    """The hoisted Swin form carries per-parameter-generation data outside the packed weight images: the 5x5 composition pred.0 o convB, the
    tap-pair products / line kernels of its border correction (handle-wide, rebuilt on the caller's stream before the lanes fork) and the E[t]
    border-class tables (per plan).  Load a second set of weights into the SAME handle: the next call must follow it (two lanes: B = 2)."""
    import diffusiondepth_amd as dda
    from oracle import ddim_oracle as O
    be = dda.HipDenoiser(variant="swin")
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    B, h, w, ch, cw, T = 2, 9, 40, 5, 20, 2
    inp = synth.make_inputs(470, B, h, w, (ch, cw))
    for wseed in (7245, 7399, 7245):
        sd = synth.make_state_dict(wseed, "swin")
        be.load_state_dict(sd)
        ref = O.ddim_loop(sd, inp["x_T"], inp["cond"], T, "swin")
        for prec in ("f16", "bf16"):
            x0 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), T, prec).cpu().numpy()
            e = U.maxabs(x0, ref) / float(np.abs(ref).max())
            U.record("swin_hoist_reload", wseed=wseed, prec=prec, latent_rel=e)
            assert e < LATENT_TOL[prec], (wseed, prec, e)


def test_conv3_without_hoisting_the_condition_term(U, golden, cases):
    """A/B switch of conv3: hoist_cond (conv3(cond) + conv3(E[t]) taken out of the loop by linearity).  Both forms must match the reference."""
    from oracle import ddim_oracle as O
    c, g = cases["loop_res"], golden("loop_res")
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    sd = U.sd_for(c)
    rag = synth.make_inputs(55, 2, 9, 33)
    ref_rag = O.ddim_loop(sd, rag["x_T"], rag["cond"], 3)
    try:
        for hoist in (0, 1):
            be.set_option("hoist_cond", hoist)
            for prec in ("fp32", "bf16", "f16x3"):
                x0 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), 20, prec).cpu().numpy()
                ref = g["x0_T20"]
                e, scale = U.maxabs(x0, ref), float(np.abs(ref).max())
                xr = be.denoise(U.cu(rag["x_T"]), U.cu(rag["cond"]), 3, prec).cpu().numpy()
                er, sr = U.maxabs(xr, ref_rag), float(np.abs(ref_rag).max())
                U.record("hoist_ab", hoist=hoist, prec=prec, latent_maxabs=e, latent_scale=scale, ragged_maxabs=er, ragged_scale=sr)
                assert e < LATENT_TOL[prec] * scale and er < LATENT_TOL[prec] * sr
    finally:
        be.set_option("hoist_cond", 0)
