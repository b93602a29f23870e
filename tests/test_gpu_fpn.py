"""GPU (`-m gpu`): the condition-aggregation FPN of the Res head (dd_condition, reference
src/model/head/ddim_depth_estimate_res.py:56-84,108-118) through the C ABI against
  (1) golden vectors minted from the reference head's own modules (fpn_odd.npz: odd pyramid, both adaptive_avg_pool2d
      size fixes active; head_res.npz: even pyramid),
  (2) the fp64 NumPy oracle on other seeded shapes (ragged tiles, batch > 1),
  (3) size-independent properties: cond handed over inside the handle == cond converted from the exported tensor
      (bit-identical x_0), batch consistency, loud errors.

Tolerances: fp32 mode 2e-5 x max|cond| (fp32 round-off class: BatchNorm is folded into the weights, so the rounding
order differs from conv -> BN); bf16 / f16 operand modes 2e-2 / 3e-3 x max|cond| (K up to 4608 products per output).
"""
import ctypes

import numpy as np
import pytest
import torch

from diffusiondepth_amd import synth

pytestmark = pytest.mark.gpu

COND_TOL = {"fp32": 2e-5, "f16": 3e-3, "bf16": 2e-2, "f16x3": 2e-5, "f16r": 2e-5}      # f16x3 / f16r: the pyramid on f16-PAIR operands (split-f16 kernels, fp32 tensors): the fp32 mode's bound


@pytest.fixture(scope="module")
def U():
    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a HIP device: the product has no CPU fallback")
    import gpu_util
    gpu_util.KVER = 2
    return gpu_util


@pytest.fixture(scope="module")
def be(U, cases):
    import diffusiondepth_amd as dda
    c = cases["head_res"]
    sd = synth.make_state_dict(c["wseed"], "res", c["decoder_gain"], c["decoder_log_scale"])
    sd.update(synth.make_fpn_state_dict(c["fseed"]))
    b = dda.HipDenoiser()
    b.load_state_dict(sd)
    b.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    yield b
    b.close()


@pytest.mark.parametrize("prec", ["fp32", "bf16", "f16", "f16x3", "f16r"])
def test_fpn_odd_pyramid_matches_reference_golden(U, be, golden, cases, prec):
    c, g = cases["fpn_odd"], golden("fpn_odd")
    fp = [U.cu(f) for f in synth.make_backbone_features(c["iseed"], c["B"], c["H"], c["W"])]
    x = be.condition(fp, prec).cpu().numpy()
    assert list(x.shape) == list(g["shape"])
    scale = float(np.abs(g["cond_ch0_8"]).max())
    e = U.maxabs(x[:, :8], g["cond_ch0_8"])
    esum = float(np.abs(x.astype(np.float64).sum(axis=(0, 2, 3)) - g["cond_chan_sum"]).max() / np.abs(g["cond_chan_sum"]).max())
    U.record("fpn_odd", prec=prec, maxabs=e, scale=scale, chan_sum_rel=esum)
    assert e <= COND_TOL[prec] * scale
    assert esum <= (1e-5 if prec in ("fp32", "f16x3", "f16r") else 5e-3)
    if prec in ("f16x3", "f16r"):
        assert be.counter("cond_split_ok") & 1                     # ... and it WAS the split-f16 kernels: the fp32-operand route gives other bits
        be.set_option("cond_split", 0)
        x32 = be.condition(fp, prec).cpu().numpy()
        be.set_option("cond_split", 1)
        assert np.array_equal(x32, be.condition(fp, "fp32").cpu().numpy()) and not np.array_equal(x32, x)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fpn_even_pyramid_matches_head_golden(U, be, golden, cases, prec):
    c, g = cases["head_res"], golden("head_res")
    fp = [U.cu(f) for f in synth.make_backbone_features(c["iseed"], c["B"], c["H"], c["W"])]
    x = be.condition(fp, prec).cpu().numpy()
    scale = float(np.abs(g["cond_ch0_4"]).max())
    e = U.maxabs(x[:, :4], g["cond_ch0_4"])
    U.record("fpn_even", prec=prec, maxabs=e, scale=scale)
    assert e <= COND_TOL[prec] * scale
    assert abs(float(x.astype(np.float64).sum()) - float(g["cond_sum"][0])) <= (1e-5 if prec == "fp32" else 5e-3) * abs(float(g["cond_sum"][0]))


@pytest.mark.parametrize("B,H,W", [(2, 50, 70), (1, 130, 34), (3, 16, 16)])
def test_fpn_matches_oracle_ragged(U, be, cases, B, H, W):
    """Ragged tiles (sizes not multiples of 8x32), 1-pixel-high top level, batch > 1; fp32 mode vs the fp64 oracle."""
    from oracle import ddim_oracle as O
    fsd = synth.make_fpn_state_dict(cases["head_res"]["fseed"])
    feats = synth.make_backbone_features(77 + B, B, H, W)
    ref = O.fpn_aggregate(fsd, feats)
    x = be.condition([U.cu(f) for f in feats], "fp32").cpu().numpy()
    scale = float(np.abs(ref).max())
    e = U.maxabs(x, ref)
    U.record("fpn_ragged", B=B, H=H, W=W, maxabs=e, scale=scale)
    assert e <= COND_TOL["fp32"] * scale
    es = U.maxabs(be.condition([U.cu(f) for f in feats], "f16x3").cpu().numpy(), ref)       # the same shapes on the split-f16 kernels
    U.record("fpn_ragged_split", B=B, H=H, W=W, maxabs=es, scale=scale)
    assert es <= COND_TOL["f16x3"] * scale


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_handle_resident_cond_equals_exported_cond(U, be, cases, prec):
    """denoise(x_T, <tensor returned by condition()>) uses the map left inside the handle (cond == NULL at the ABI);
    handing the same values over as a fresh tensor goes through the NCHW conversion.  x_0 must be bit-identical."""
    c = cases["head_res"]
    B, H, W = 2, 48, 80
    fp = [U.cu(f) for f in synth.make_backbone_features(5, B, H, W)]
    h, w = fp[0].shape[2], fp[0].shape[3]
    x_T = U.cu(synth.make_inputs(9, B, h, w)["x_T"])
    cond = be.condition(fp, prec)
    assert be._cond_arg(cond, prec) is None
    a = be.denoise(x_T, cond, c["T"], prec).cpu().numpy()
    cond2 = cond.clone()
    assert be._cond_arg(cond2, prec) is not None
    b = be.denoise(x_T, cond2, c["T"], prec).cpu().numpy()
    assert np.array_equal(a, b)
    # in-place modification of the returned tensor invalidates the shortcut
    cond = be.condition(fp, prec)
    cond.mul_(1.0)
    assert be._cond_arg(cond, prec) is not None
    # eps path as well
    cond = be.condition(fp, prec)
    t = torch.tensor([10, 500], device="cuda")
    e1 = be.denoise_once(x_T, t, cond, prec).cpu().numpy()
    e2 = be.denoise_once(x_T, t, cond.clone(), prec).cpu().numpy()
    assert np.array_equal(e1, e2)


def test_fpn_batch_consistency(U, be):
    feats = synth.make_backbone_features(3, 3, 40, 72)
    full = be.condition([U.cu(f) for f in feats], "bf16").cpu().numpy()
    one = be.condition([U.cu(f[1:2]) for f in feats], "bf16").cpu().numpy()
    assert np.array_equal(full[1:2], one)


def test_fpn_errors_are_loud(U, be, cases):
    import diffusiondepth_amd as dda
    lib = be._lib
    x = torch.zeros(1, 16, 8, 8, device="cuda")
    out = torch.empty_like(x)
    # cond == NULL without a preceding dd_condition of that shape
    rc = lib.dd_denoise(be._h, x.data_ptr(), None, out.data_ptr(), 1, 8, 8, 8, 8, 5, 1, None)
    assert rc != 0 and b"dd_condition" in lib.dd_last_error(be._h)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        be.condition([torch.zeros(1, 64 << i, 8 >> i, 8 >> i) for i in range(4)], "fp32")
    with pytest.raises(ValueError):
        be.condition([torch.zeros(1, 64, 8, 8, device="cuda")] * 4, "fp32")
    with pytest.raises(RuntimeError, match="precision"):
        be.condition([torch.zeros(1, 64 << i, 8 >> i, 8 >> i, device="cuda") for i in range(4)], "naive_fp32")
    # a handle without FPN weights refuses
    b2 = dda.HipDenoiser()
    b2.load_state_dict(synth.make_state_dict(cases["head_res"]["wseed"], "res"))
    with pytest.raises(RuntimeError, match="conv_lateral"):
        b2.condition([torch.zeros(1, 64 << i, 8 >> i, 8 >> i, device="cuda") for i in range(4)], "fp32")
    b2.close()


@pytest.mark.parametrize("size", ["nyu", "kitti"])
def test_fpn_full_size_vs_torch_modules(U, be, cases, size):
    """BASELINE.json sizes: NYU 228x304 (odd pyramid -> pooling active) and KITTI 352x1216, fp32 mode vs the head's
    own torch modules (conv_lateral / conv_up) run by PyTorch-ROCm on the same device."""
    import diffusiondepth_amd as dda
    H, W = (228, 304) if size == "nyu" else (352, 1216)
    c = cases["head_res"]
    fsd = synth.make_fpn_state_dict(c["fseed"])
    head = dda.DDIMDepthEstimate_Res(precision="fp32", condition_backend="torch").eval()
    head.load_state_dict({k: torch.from_numpy(v) for k, v in fsd.items()}, strict=False)
    head = head.cuda()
    prev = torch.backends.cudnn.allow_tf32
    torch.backends.cudnn.allow_tf32 = False
    try:
        fp = [U.cu(f) for f in synth.make_backbone_features(21, 1, H, W)]
        with torch.no_grad():
            ref = head.aggregate_condition(fp).cpu().numpy()
    finally:
        torch.backends.cudnn.allow_tf32 = prev
    x = be.condition(fp, "fp32").cpu().numpy()
    scale = float(np.abs(ref).max())
    e = U.maxabs(x, ref)
    U.record("fpn_full", size=size, maxabs=e, scale=scale)
    assert e <= 1e-4 * scale      # two fp32 implementations with different summation orders (K up to 4608)
    xs = be.condition(fp, "f16r").cpu().numpy()              # the headline mode's pyramid: split-f16 kernels
    es = U.maxabs(xs, ref)
    U.record("fpn_full_split", size=size, maxabs=es, scale=scale)
    assert es <= 1e-4 * scale
    xb = be.condition(fp, "bf16").cpu().numpy()
    eb = U.rms(xb, ref) / float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    U.record("fpn_full_bf16", size=size, rel_rms=eb)
    assert eb <= 1e-2


@pytest.fixture(scope="module")
def be_swin(U, cases):
    import diffusiondepth_amd as dda
    c = cases["loop_swin"]
    sd = synth.make_state_dict(c["wseed"], "swin", c.get("decoder_gain", 0.05), c.get("decoder_log_scale", 0.0))
    sd.update(synth.make_fpn_state_dict(c.get("fseed", 7242), in_channels=(192, 384, 768, 1536)))
    b = dda.HipDenoiser(variant="swin")
    b.load_state_dict(sd)
    b.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    yield b
    b.close()


@pytest.mark.parametrize("prec", ["fp32", "bf16", "f16x3"])
def test_swin_fpn_matches_oracle(U, be_swin, cases, prec):
    """Swin-L pyramid widths (192..1536): lateral convs with 6..48 channel chunks; odd sizes -> pooling active."""
    from oracle import ddim_oracle as O
    fsd = synth.make_fpn_state_dict(cases["loop_swin"].get("fseed", 7242), in_channels=(192, 384, 768, 1536))
    feats = synth.make_backbone_features(55, 1, 46, 78, in_channels=(192, 384, 768, 1536))     # 23x39, 12x20, 6x10, 3x5
    ref = O.fpn_aggregate(fsd, feats)
    x = be_swin.condition([U.cu(f) for f in feats], prec).cpu().numpy()
    scale = float(np.abs(ref).max())
    e = U.maxabs(x, ref)
    U.record("fpn_swin", prec=prec, maxabs=e, scale=scale)
    assert e <= COND_TOL[prec] * scale


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_swin_handle_resident_cond_equals_exported_cond(U, be_swin, cases, prec):
    """Swin: the stride-4 condition map left in the handle is upsampled blocked -> blocked; handing the exported tensor
    over goes through the NCHW upsample kernel.  Same fp32 formula on the same values; the two kernels may contract the
    interpolation FMAs differently, so x_0 agrees to fp32 round-off (bf16: the rounded operands are the same or one ulp apart)."""
    c = cases["loop_swin"]
    feats = synth.make_backbone_features(56, 2, 40, 72, in_channels=(192, 384, 768, 1536))      # cond 20x36
    fp = [U.cu(f) for f in feats]
    h, w = 40, 72                                                                              # latent twice the condition size
    x_T = U.cu(synth.make_inputs(9, 2, h, w)["x_T"])
    cond = be_swin.condition(fp, prec)
    assert be_swin._cond_arg(cond, prec) is None
    a = be_swin.denoise(x_T, cond, 20, prec).cpu().numpy()
    b = be_swin.denoise(x_T, cond.clone(), 20, prec).cpu().numpy()
    assert U.maxabs(a, b) <= (2e-6 if prec == "fp32" else 2e-3) * float(np.abs(a).max())


@pytest.mark.parametrize("prec", ["fp32", "bf16", "f16"])
def test_concurrent_lanes_on_the_handle_resident_cond(U, be, be_swin, cases, prec):
    """dd_set_option("streams", S): the batch runs as S concurrent sub-batches on separate HIP streams (own plans / graphs; fork and join by
    events on the caller's stream).  With the condition map left in the handle by dd_condition every lane reads ITS images of the whole
    batch's map (Res: an alias into that buffer, no copy; Swin: its slice of the stride-4 map, upsampled into the lane's own buffer);
    with an explicit tensor every lane converts its slice.  Bit-identical to one stream: the images are independent."""
    for b, chans, lat in ((be, None, None), (be_swin, (192, 384, 768, 1536), (40, 72))):
        B = 3
        feats = synth.make_backbone_features(5, B, 48 if chans is None else 40, 80 if chans is None else 72, **({} if chans is None else {"in_channels": chans}))
        fp = [U.cu(f) for f in feats]
        h, w = lat if lat else (fp[0].shape[2], fp[0].shape[3])
        x_T = U.cu(synth.make_inputs(9, B, h, w)["x_T"])
        try:
            b.set_option("streams", 1)
            cond = b.condition(fp, prec)
            want = b.denoise(x_T, cond, 5, prec)
            want_explicit = b.denoise(x_T, cond.clone(), 5, prec)
            for S in (2, 3):
                b.set_option("streams", S)
                n0 = b.counter("lane_calls")
                cond = b.condition(fp, prec)
                assert b._cond_arg(cond, prec) is None
                got = b.denoise(x_T, cond, 5, prec)
                again = b.denoise(x_T, b.condition(fp, prec), 5, prec)              # replayed lane graphs, refreshed condition map
                got_explicit = b.denoise(x_T, cond.clone(), 5, prec)
                assert b.counter("lane_calls") == n0 + 3
                assert torch.equal(got, want) and torch.equal(again, want) and torch.equal(got_explicit, want_explicit), (prec, S)
        finally:
            b.set_option("streams", 1)
