"""GPU (`-m gpu`): the NLSPN refinement stage and the DCNv2 operator under it (csrc/dd_dcn.hip through the C ABI of
include/ddepth_dcn.h) against
  (1) goldens minted from the reference's own code (tests/golden/nlspn_*.npz, dcn_*.npz: the reference NLSPN class on the reference's
      DCN device code compiled for the host, tests/golden/make_golden_nlspn.py),
  (2) the fp64 NumPy oracle (oracle/dcn_oracle.py, itself pinned to both) on further seeded shapes,
  (3) the known answers of the reference's self-test (src/model/deformconv/test.py),
  (4) size-independent properties at the full KITTI resolution 352 x 1216 (identity affinities, fused loop == one operator call per
      iteration, linearity in the propagated map).
Tolerance: fp32 round-off class -- 1e-5 relative to the largest magnitude of the compared tensor (measured values are recorded in
gpurun_out/parity_report.jsonl); the arithmetic is fp32 on both sides, differences come from FMA contraction and summation order.
(The file name sorts last on purpose: these kernels are the newest in the round.)"""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def U():
    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a HIP device: the product has no CPU fallback")
    import gpu_util
    return gpu_util


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-12))


def _args(g=None, **kw):
    d = dict(prop_time=18, affinity="TGASS", affinity_gamma=0.5, conf_prop=True, preserve_input=False, legacy=False)
    if g is not None:
        B, H, W, ch_g, k_f, T, cp, pi, lg = [int(v) for v in g["meta"]]
        d.update(prop_time=T, affinity=str(g["affinity"]), conf_prop=bool(cp), preserve_input=bool(pi), legacy=bool(lg))
    d.update(kw)
    return types.SimpleNamespace(**d)


def _module(g, U):
    from diffusiondepth_amd.nlspn import NLSPN
    B, H, W, ch_g, k_f, T, cp, pi, lg = [int(v) for v in g["meta"]]
    m = NLSPN(_args(g), ch_g, 1, 3, k_f).cuda()
    with torch.no_grad():
        m.conv_offset_aff.weight.copy_(U.cu(g["conv_weight"]))
        m.conv_offset_aff.bias.copy_(U.cu(g["conv_bias"]))
    return m


@pytest.mark.parametrize("name", ["groups", "dg_stride", "k1"])
def test_dcn_forward_backward_match_reference_goldens(U, golden, name):
    from diffusiondepth_amd import dcn
    g = golden("dcn_" + name)
    sh, sw, ph, pw, dh, dw, grp, dg, step = [int(v) for v in g["meta"]]
    x, w, b, off, m, go = (U.cu(g[k]) for k in ("input", "weight", "bias", "offset", "mask", "grad_out"))
    kh, kw = w.shape[2:]
    y = dcn.modulated_deform_conv_forward(x, w, b, off, m, kh, kw, sh, sw, ph, pw, dh, dw, grp, dg, step)
    errs = {"out": rel(y, g["out"])}
    grads = dcn.modulated_deform_conv_backward(x, w, b, off, m, go, kh, kw, sh, sw, ph, pw, dh, dw, grp, dg, step)
    for k, v in zip(("g_input", "g_offset", "g_mask", "g_weight", "g_bias"), grads):
        errs[k] = rel(v, g[k])
    U.record("dcn_" + name, **errs)
    assert max(errs.values()) < TOL, errs
    # outputs nobody needs are skipped, the rest is unchanged
    part = dcn.modulated_deform_conv_backward(x, w, b, off, m, go, kh, kw, sh, sw, ph, pw, dh, dw, grp, dg, step,
                                              needs=(False, True, True, False, False))
    assert part[0] is None and part[3] is None and part[4] is None
    assert torch.equal(part[1], grads[1]) and torch.equal(part[2], grads[2])


@pytest.mark.parametrize("shape", [(2, 6, 4, 11, 13, 3, 3, (1, 2), (1, 0), (2, 1), 2, 3), (1, 3, 5, 17, 9, 5, 3, (1, 1), (2, 1), (1, 1), 1, 1),
                                   (3, 1, 1, 33, 70, 3, 3, (1, 1), (1, 1), (1, 1), 1, 1)], ids=["groups_dg_asym", "k5x3", "nlspn_like"])
def test_dcn_matches_oracle_on_seeded_shapes(U, shape):
    """Includes pad_h != pad_w (the reference's col2im uses pad_h for both, reproduced) and non-square kernels."""
    from diffusiondepth_amd import dcn
    from oracle import dcn_oracle as O
    B, C, Co, H, W, kh, kw, st, pd, dl, grp, dg = shape
    rs = np.random.RandomState(abs(hash(shape)) % 2**31)
    Ho, Wo = O.out_size(H, W, kh, kw, st, pd, dl)
    x = rs.standard_normal((B, C, H, W)).astype(np.float32)
    w = rs.standard_normal((Co, C // grp, kh, kw)).astype(np.float32)
    b = rs.standard_normal(Co).astype(np.float32)
    off = (3.0 * rs.standard_normal((B, dg * 2 * kh * kw, Ho, Wo))).astype(np.float32)
    m = rs.uniform(0, 2, (B, dg * kh * kw, Ho, Wo)).astype(np.float32)
    go = rs.standard_normal((B, Co, Ho, Wo)).astype(np.float32)
    a = [U.cu(t) for t in (x, w, b, off, m)]
    y = dcn.modulated_deform_conv_forward(*a, kh, kw, st[0], st[1], pd[0], pd[1], dl[0], dl[1], grp, dg, 64)
    errs = {"out": rel(y, O.mdcn_forward(x, w, b, off, m, st, pd, dl, grp, dg))}
    grads = dcn.modulated_deform_conv_backward(*a, U.cu(go), kh, kw, st[0], st[1], pd[0], pd[1], dl[0], dl[1], grp, dg, 64)
    for k, v, r in zip(("g_input", "g_offset", "g_mask", "g_weight", "g_bias"), grads, O.mdcn_backward(x, w, b, off, m, go, st, pd, dl, grp, dg)):
        errs[k] = rel(v, r)
    U.record("dcn_oracle_%dx%d" % (H, W), **errs)
    assert max(errs.values()) < TOL, errs


def test_dcn_reference_selftest_known_answers_and_errors(U):
    """src/model/deformconv/test.py: zero offset == nn.Conv2d (:104), identity kernel (:172-174), im2col_step invariance (:244);
    argument errors raise RuntimeError like AT_ASSERTM (modulated_deform_conv_cuda.cu:39-72)."""
    from diffusiondepth_amd import dcn
    torch.manual_seed(3)
    x = torch.randn(2, 4, 4, 4, device="cuda")
    w = torch.randn(4, 2, 3, 3, device="cuda")
    b = torch.randn(4, device="cuda")
    off = torch.zeros(2, 18, 4, 4, device="cuda")
    ones = torch.ones(2, 9, 4, 4, device="cuda")
    f = lambda x_, w_, b_, o_, m_, step=1: dcn.modulated_deform_conv_forward(x_, w_, b_, o_, m_, 3, 3, 1, 1, 1, 1, 1, 1, 2, 1, step)
    assert float((f(x, w, b, off, ones) - torch.nn.functional.conv2d(x, w, b, padding=1, groups=2)).abs().max()) < 1e-5
    wi = torch.zeros_like(w)
    for q in range(4):
        wi[q, q % 2, 1, 1] = 1.0
    assert float((2 * f(x, wi, torch.zeros_like(b), off, 0.5 * ones) - x).abs().max()) < 1e-6
    off_r, m_r = torch.randn(2, 18, 4, 4, device="cuda"), torch.rand(2, 9, 4, 4, device="cuda")
    assert torch.equal(f(x, w, b, off_r, m_r, 1), f(x, w, b, off_r, m_r, 2))
    x3 = torch.randn(3, 4, 4, 4, device="cuda")
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        f(x3, w, b, torch.zeros(3, 18, 4, 4, device="cuda"), torch.ones(3, 9, 4, 4, device="cuda"), 2)
    with pytest.raises(RuntimeError, match="contiguous"):
        f(x.transpose(2, 3), w, b, off, ones)
    with pytest.raises(RuntimeError, match="CPU"):
        f(x.cpu(), w, b, off, ones)
    with pytest.raises(RuntimeError, match="kernel channels wont match"):
        dcn.modulated_deform_conv_forward(x, w, b, off, ones, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1)


@pytest.mark.parametrize("name", ["tgass", "preserve", "as_noconf", "tc_legacy", "k5"])
def test_nlspn_fused_path_matches_reference_goldens(U, golden, name):
    g = golden("nlspn_" + name)
    m = _module(g, U).eval()
    cp = bool(int(g["meta"][6]))
    with torch.no_grad():
        y, y_inter, offset, aff, gamma = m(U.cu(g["feat_init"]), U.cu(g["guidance"]), U.cu(g["confidence"]) if cp else None, U.cu(g["feat_fix"]))
    scale = np.abs(g["y_inter"]).max()
    errs = {"offset_abs": U.maxabs(offset.cpu().numpy(), g["offset"]), "aff_abs": U.maxabs(aff.cpu().numpy(), g["aff"]),
            "y": U.maxabs(y.cpu().numpy(), g["y"]) / scale,
            "y_inter": U.maxabs(torch.stack(y_inter).cpu().numpy(), g["y_inter"]) / scale}
    U.record("nlspn_" + name, **errs)
    assert errs["offset_abs"] < 2e-5 and errs["aff_abs"] < 1e-5 and errs["y"] < TOL and errs["y_inter"] < TOL, errs
    assert len(y_inter) == int(g["meta"][5]) and torch.equal(y_inter[-1], y)
    assert float(gamma) == float(g["aff_const"][0])


def test_nlspn_training_path_matches_reference_autograd(U, golden):
    """Gradients of sum(y * G) through prop_time = 18 iterations + the confidence sampling, as minted from the reference's autograd
    (tgass golden).  Tolerance 1e-4 relative: 18 chained fp32 atomically-accumulated scatters."""
    g = golden("nlspn_tgass")
    m = _module(g, U).train()
    feat, guide, conf = (U.cu(g[k]).requires_grad_(True) for k in ("feat_init", "guidance", "confidence"))
    y, y_inter, offset, aff, _ = m(feat, guide, conf, U.cu(g["feat_fix"]))
    scale = np.abs(g["y_inter"]).max()
    assert U.maxabs(y.detach().cpu().numpy(), g["y"]) / scale < TOL          # the autograd path computes the same forward
    (y * U.cu(g["grad_y"])).sum().backward()
    errs = {"g_feat_init": rel(feat.grad, g["g_feat_init"]), "g_guidance": rel(guide.grad, g["g_guidance"]),
            "g_confidence": rel(conf.grad, g["g_confidence"]), "g_conv_weight": rel(m.conv_offset_aff.weight.grad, g["g_conv_weight"]),
            "g_conv_bias": rel(m.conv_offset_aff.bias.grad, g["g_conv_bias"]),
            "g_aff_scale_const": rel(m.aff_scale_const.grad, g["g_aff_scale_const"])}
    U.record("nlspn_train_tgass", **errs)
    assert max(errs.values()) < 1e-4, errs
    assert m.w.grad is None and m.b.grad is None


def test_nlspn_full_size_properties(U):
    """KITTI 352 x 1216 (BASELINE config 5's refinement size), B = 2."""
    from diffusiondepth_amd import dcn
    from diffusiondepth_amd.nlspn import NLSPN
    B, H, W, T = 2, 352, 1216, 18
    gen = torch.Generator(device="cuda").manual_seed(7)
    feat = 10 * torch.rand(B, 1, H, W, device="cuda", generator=gen)
    m = NLSPN(_args(), 8, 1, 3, 3).cuda().eval()
    w1, b0 = m.w, m.b
    # (a) identity: zero offsets, reference affinity 1 -> every iteration returns its input exactly
    off0 = torch.zeros(B, 18, H, W, device="cuda")
    aff0 = torch.zeros(B, 9, H, W, device="cuda")
    aff0[:, 4] = 1.0
    feats = dcn.nlspn_propagate(feat, off0, aff0, None, w1, b0, 3, T, False)
    assert torch.equal(feats[-1], feat) and torch.equal(feats[0], feat)
    # (b) the module with its zero-initialised conv (nlspnmodel.py:55-56) is therefore the identity, whatever the guidance
    with torch.no_grad():
        y, y_inter, offset, aff, _ = m(feat, torch.randn(B, 8, H, W, device="cuda", generator=gen), torch.rand(B, 1, H, W, device="cuda", generator=gen))
    assert torch.equal(y, feat) and float(offset.abs().max()) == 0.0 and torch.equal(aff, aff0)
    # (c) random offsets / affinities: fused loop == one DCNv2 operator call per iteration (the reference's formulation)
    off = 2.5 * torch.randn(B, 18, H, W, device="cuda", generator=gen)
    a = 0.1 * torch.randn(B, 9, H, W, device="cuda", generator=gen)
    a[:, 4] = 1.0 - (a.sum(1) - a[:, 4])
    fix = torch.where(torch.rand(B, 1, H, W, device="cuda", generator=gen) < 0.05, 20 * torch.rand(B, 1, H, W, device="cuda", generator=gen),
                      torch.zeros(B, 1, H, W, device="cuda"))
    for preserve in (False, True):
        fused = dcn.nlspn_propagate(feat, off, a, fix if preserve else None, w1, b0, 3, 6, preserve)
        cur = feat
        mask = (fix > 0).float()
        for k in range(6):
            if preserve:
                cur = (1.0 - mask) * cur + mask * fix
            cur = dcn.modulated_deform_conv_forward(cur, w1, b0, off, a, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64)
            assert float((fused[k] - cur).abs().max()) <= 1e-5 * float(cur.abs().max()), (preserve, k)
    # (d) linearity in the propagated map (offsets / affinities fixed, bias 0): P(2 f + g) == 2 P(f) + P(g)
    f2 = torch.randn(B, 1, H, W, device="cuda", generator=gen)
    p = lambda t: dcn.nlspn_propagate(t, off, a, None, w1, b0, 3, 6, False)[-1]
    lhs, rhs = p(2 * feat + f2), 2 * p(feat) + p(f2)
    err = float((lhs - rhs).abs().max() / rhs.abs().max())
    U.record("nlspn_full_size_linearity", err=err)
    assert err < 1e-5
    # (f) the guidance convolution evaluated inside the affinity kernel (dd_nlspn_guided_offset_affinity) == torch conv + dd_nlspn_offset_affinity
    m2 = NLSPN(_args(), 8, 1, 3, 3).cuda().eval()
    with torch.no_grad():
        m2.conv_offset_aff.weight.copy_(0.1 * torch.randn(24, 8, 3, 3, device="cuda", generator=gen))
        m2.conv_offset_aff.bias.copy_(0.3 * torch.randn(24, device="cuda", generator=gen))
        guide, conf = 2 * torch.randn(B, 8, H, W, device="cuda", generator=gen), torch.rand(B, 1, H, W, device="cuda", generator=gen)
        for legacy in (False, True):
            m2.args.legacy = legacy
            m2.fuse_guidance_conv = True
            _, _, o_f, a_f, _ = m2(feat, guide, conf)
            m2.fuse_guidance_conv = False
            _, _, o_u, a_u, _ = m2(feat, guide, conf)
            # reference point for the convolution: fp64 on the host (MIOpen picks a Winograd fp32 kernel for the unfused path, which is
            # the LESS accurate of the two: its distance is recorded, the fused kernel is held to the tight bound)
            oa64 = torch.nn.functional.conv2d(guide[:1].double().cpu(), m2.conv_offset_aff.weight.double().cpu(),
                                              m2.conv_offset_aff.bias.double().cpu(), padding=1).float().cuda()
            o_r, a_r = dcn.nlspn_offset_affinity(oa64, conf[:1], m2.aff_scale_const, m2.w_conf, m2.b, 3, "TGASS", True, legacy)
            eo, ea = float((o_f[:1] - o_r).abs().max()), float((a_f[:1] - a_r).abs().max())
            eo_m, ea_m = float((o_u[:1] - o_r).abs().max()), float((a_u[:1] - a_r).abs().max())
            U.record("nlspn_guided_conv", legacy=int(legacy), fused_offset_abs=eo, fused_aff_abs=ea, miopen_offset_abs=eo_m, miopen_aff_abs=ea_m,
                     offset_max=float(o_r.abs().max()))
            # offsets: fp32 round-off of values up to ~11.  Affinities: the confidence is sampled at (float)w + offset with w up to 1215,
            # i.e. on a 1.2e-4 grid (fp32 ulp at 1024..2048) -- in the reference too (modulated_deform_im2col_cuda.cuh:177-178) -- so a
            # 1e-6 difference in an offset can move a sample by one grid step: |d aff| <= 1.2e-4 * |grad conf| * 0.25 per neighbour
            assert eo < 4e-6 and ea < 2e-4, (legacy, eo, ea)
            assert float((o_f - o_u).abs().max()) < 5e-5 and float((a_f - a_u).abs().max()) < 2e-4       # fused vs MIOpen-based path, whole batch
    with pytest.raises(RuntimeError, match="built for ch_g 8"):
        dcn.nlspn_guided_offset_affinity(torch.zeros(1, 4, 8, 8, device="cuda"), torch.zeros(24, 4, 3, 3, device="cuda"), torch.zeros(24, device="cuda"),
                                         None, m.aff_scale_const, m.w_conf, m.b, 3, 3, "TGASS", False, False)
    # (e) an odd width takes the one-pixel-per-lane kernel: same numbers as the 4-pixel kernel on the common columns' interior
    Wc = W - 3
    crop = lambda t: t[..., :Wc].contiguous()
    fo = dcn.nlspn_propagate(crop(feat), torch.zeros(B, 18, H, Wc, device="cuda"), crop(aff0), None, w1, b0, 3, 2, False)
    assert torch.equal(fo[-1], crop(feat))
