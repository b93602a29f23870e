"""CPU: the NLSPN / DCNv2 oracle (oracle/dcn_oracle.py) is pinned to the reference:
  (1) against the reference's OWN device code compiled for the host (oracle/_ref/libref_dcn.so, built by oracle/ref_dcn/build_ref.py
      from /root/reference/src/model/deformconv/src/cuda/modulated_deform_im2col_cuda.cuh) on seeded random cases;
  (2) against goldens minted by running the reference's NLSPN class / DCN kernels on CPU (tests/golden/make_golden_nlspn.py);
  (3) against the known answers of the reference's own self-test (src/model/deformconv/test.py): zero offsets == nn.Conv2d
      (check_mdconv_zero_offset, :68-110), identity kernel (check_mdconv_zero_offset_identify, :140-177), im2col_step
      invariance (check_mdconv_im2col_step_forward, :210-250);
plus the host-side mirror (constructor contract, state_dict names, loud failure without a GPU) and the C ABI of include/ddepth_dcn.h.
Tolerances: fp32 round-off class, 2e-6 relative to the largest magnitude of the compared tensor (the golden / reference side is fp32)."""
import os
import re
import types

import numpy as np
import pytest
import torch

from oracle import dcn_oracle as O, dcn_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-6
NLSPN_CASES = ["tgass", "preserve", "as_noconf", "tc_legacy", "k5"]


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-12))


def _rand_case(rs, B, C, Co, H, W, kh, kw, st, pd, dl, grp, dg):
    Ho, Wo = O.out_size(H, W, kh, kw, st, pd, dl)
    return dict(x=rs.standard_normal((B, C, H, W)).astype(np.float32), w=rs.standard_normal((Co, C // grp, kh, kw)).astype(np.float32),
                b=rs.standard_normal(Co).astype(np.float32), off=(2.5 * rs.standard_normal((B, dg * 2 * kh * kw, Ho, Wo))).astype(np.float32),
                m=rs.uniform(0, 2, (B, dg * kh * kw, Ho, Wo)).astype(np.float32), go=rs.standard_normal((B, Co, Ho, Wo)).astype(np.float32))


SHAPES = [(2, 4, 4, 5, 6, 3, 3, (1, 1), (1, 1), (1, 1), 2, 1), (2, 4, 6, 7, 5, 3, 3, (2, 1), (1, 1), (1, 2), 1, 2),
          (1, 1, 1, 9, 11, 3, 3, (1, 1), (1, 1), (1, 1), 1, 1), (2, 1, 1, 6, 7, 1, 1, (1, 1), (0, 0), (1, 1), 1, 1),
          (1, 2, 2, 8, 8, 5, 5, (1, 1), (2, 2), (1, 1), 1, 1)]


@pytest.mark.skipif(not dcn_ref.available(), reason="neither oracle/_ref/libref_dcn.so nor /root/reference present")
@pytest.mark.parametrize("shape", SHAPES, ids=[f"case{i}" for i in range(len(SHAPES))])
def test_oracle_matches_reference_device_code(shape):
    B, C, Co, H, W, kh, kw, st, pd, dl, grp, dg = shape
    c = _rand_case(np.random.RandomState(hash(shape) % 2**31), *shape)
    y_ref = dcn_ref.forward(c["x"], c["w"], c["b"], c["off"], c["m"], st, pd, dl, grp, dg, im2col_step=1)
    assert rel(O.mdcn_forward(c["x"], c["w"], c["b"], c["off"], c["m"], st, pd, dl, grp, dg), y_ref) < TOL
    g_ref = dcn_ref.backward(c["x"], c["w"], c["b"], c["off"], c["m"], c["go"], st, pd, dl, grp, dg)
    g_or = O.mdcn_backward(c["x"], c["w"], c["b"], c["off"], c["m"], c["go"], st, pd, dl, grp, dg)
    for name, a, r in zip(("input", "offset", "mask", "weight", "bias"), g_or, g_ref):
        assert rel(a, r) < TOL, name


@pytest.mark.skipif(not dcn_ref.available(), reason="neither oracle/_ref/libref_dcn.so nor /root/reference present")
def test_reference_col2im_uses_pad_h_for_both_paddings():
    """The launcher slip (modulated_deform_im2col_cuda.cuh:372): only visible when pad_h != pad_w; the oracle reproduces it."""
    shape = (1, 2, 2, 6, 9, 3, 3, (1, 1), (1, 2), (1, 1), 1, 1)
    c = _rand_case(np.random.RandomState(5), *shape)
    g_ref = dcn_ref.backward(c["x"], c["w"], c["b"], c["off"], c["m"], c["go"], shape[7], shape[8], shape[9], 1, 1)[0]
    g_slip = O.mdcn_backward(c["x"], c["w"], c["b"], c["off"], c["m"], c["go"], shape[7], shape[8], shape[9], 1, 1, pad_w_slip=True)[0]
    g_math = O.mdcn_backward(c["x"], c["w"], c["b"], c["off"], c["m"], c["go"], shape[7], shape[8], shape[9], 1, 1, pad_w_slip=False)[0]
    assert rel(g_slip, g_ref) < TOL
    assert rel(g_math, g_ref) > 1e-2


@pytest.mark.parametrize("name", ["groups", "dg_stride", "k1"])
def test_oracle_matches_dcn_goldens(golden, name):
    g = golden("dcn_" + name)
    sh, sw, ph, pw, dh, dw, grp, dg, step = [int(v) for v in g["meta"]]
    a = (g["input"], g["weight"], g["bias"], g["offset"], g["mask"])
    assert rel(O.mdcn_forward(*a, (sh, sw), (ph, pw), (dh, dw), grp, dg, step), g["out"]) < TOL
    grads = O.mdcn_backward(*a, g["grad_out"], (sh, sw), (ph, pw), (dh, dw), grp, dg)
    for k, v in zip(("g_input", "g_offset", "g_mask", "g_weight", "g_bias"), grads):
        assert rel(v, g[k]) < TOL, k


@pytest.mark.parametrize("name", NLSPN_CASES)
def test_oracle_matches_nlspn_goldens(golden, name):
    g = golden("nlspn_" + name)
    B, H, W, ch_g, k_f, T, cp, pi, lg = [int(v) for v in g["meta"]]
    off, aff = O.nlspn_offset_affinity(g["offset_aff"], g["confidence"] if cp else None, g["aff_const"][0], str(g["affinity"]), k_f,
                                       bool(cp), bool(lg))
    assert np.abs(off - g["offset"]).max() < 1e-6 and np.abs(aff - g["aff"]).max() < 1e-6
    y, ys = O.nlspn_propagate(g["feat_init"], off, aff, g["feat_fix"], T, bool(pi), k_f)
    scale = np.abs(g["y_inter"]).max()
    assert np.abs(y - g["y"]).max() < TOL * scale and np.abs(np.stack(ys) - g["y_inter"]).max() < TOL * scale
    # the conv in front of the stage is a plain Conv2d: its stored output is what the fixtures' weights give
    oa = torch.nn.functional.conv2d(torch.from_numpy(g["guidance"]), torch.from_numpy(g["conv_weight"]), torch.from_numpy(g["conv_bias"]), padding=1)
    assert np.abs(oa.numpy() - g["offset_aff"]).max() < 1e-5


def test_reference_selftest_known_answers():
    """src/model/deformconv/test.py: N, inC, inH, inW = 2, 4, 4, 4; outC 4; 3x3; groups 2."""
    rs = np.random.RandomState(3)
    x = rs.standard_normal((2, 4, 4, 4)).astype(np.float32)
    w = rs.standard_normal((4, 2, 3, 3)).astype(np.float32)
    b = rs.standard_normal(4).astype(np.float32)
    off = np.zeros((2, 18, 4, 4), np.float32)
    ones = np.ones((2, 9, 4, 4), np.float32)                       # sigmoid(0) * 2 (test.py:99-100)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1, groups=2).numpy()
    assert np.abs(O.mdcn_forward(x, w, b, off, ones, pad=(1, 1), group=2) - ref).max() < 1e-5            # "mdconv zero offset", test.py:104
    wi = np.zeros_like(w)
    for q in range(4):                                              # conv_identify (test.py:24-36)
        wi[q, q % 2, 1, 1] = 1.0
    half = np.full((2, 9, 4, 4), 0.5, np.float32)                   # sigmoid(0) (test.py:170)
    assert np.abs(2 * O.mdcn_forward(x, wi, np.zeros(4, np.float32), off, half, pad=(1, 1), group=2) - x).max() < 1e-6   # test.py:172-174
    off_r = rs.standard_normal((2, 18, 4, 4)).astype(np.float32)
    m_r = rs.uniform(0, 1, (2, 9, 4, 4)).astype(np.float32)
    y1 = O.mdcn_forward(x, w, b, off_r, m_r, pad=(1, 1), group=2, im2col_step=1)
    y2 = O.mdcn_forward(x, w, b, off_r, m_r, pad=(1, 1), group=2, im2col_step=2)
    assert np.array_equal(y1, y2)                                   # "mdconv im2col_step forward", test.py:244
    with pytest.raises(ValueError):
        O.mdcn_forward(np.concatenate([x, x[:1]]), w, b, np.zeros((3, 18, 4, 4), np.float32), np.ones((3, 9, 4, 4), np.float32), pad=(1, 1),
                       group=2, im2col_step=2)                       # batch(3) % im2col_step(2)


def test_library_exports_every_dcn_symbol():
    from diffusiondepth_amd import dcn
    hdr = open(os.path.join(ROOT, "include", "ddepth_dcn.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(dd_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(dcn.ABI_SYMBOLS), declared ^ set(dcn.ABI_SYMBOLS)
    lib = dcn._lib()
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.dd_dcn_last_error() == b""


def test_ctypes_signatures_match_the_header_prototypes():
    """Arity of every bound function == the parameter count of its prototype in include/*.h (a wrong argtypes list only shows on the GPU
    box otherwise), and argument validation answers before any HIP call (NULL pointers -> DD_ERR_INVALID_ARG + message, no GPU needed)."""
    import diffusiondepth_amd as dda
    from diffusiondepth_amd import dcn
    lib = dcn._lib()
    dda.load_library()
    for hdr in ("ddepth.h", "ddepth_dcn.h"):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", hdr)).read(), flags=re.S)
        for name, params in re.findall(r"(?:int|const char\*)\s+(dd_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
            n = 0 if params.strip() in ("", "void") else params.count(",") + 1
            fn = getattr(lib, name)
            assert fn.argtypes is not None and len(fn.argtypes) == n, (name, n, len(fn.argtypes or ()))
    assert lib.dd_nlspn_guided_offset_affinity(*([None] * 9), 1, 8, 4, 4, 3, 3, 3, 1, 0, None) == 1
    assert b"null tensor pointer" in lib.dd_dcn_last_error()
    assert lib.dd_dcn_forward(*([None] * 6), *([1] * 16), None) == 1 and lib.dd_dcn_backward(*([None] * 11), *([1] * 16), None) == 1
    assert lib.dd_nlspn_propagate(*([None] * 8), 1, 4, 4, 3, 18, 0, None) == 1 and lib.dd_nlspn_offset_affinity(*([None] * 7), 1, 4, 4, 3, 3, 1, 0, None) == 1


def _args(**kw):
    d = dict(prop_time=18, affinity="TGASS", affinity_gamma=0.5, conf_prop=True, preserve_input=False, legacy=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_nlspn_mirror_contract():
    from diffusiondepth_amd.nlspn import NLSPN
    m = NLSPN(_args(), 8, 1, 3, 3)
    # parameter names / shapes of the reference module (src/model/nlspnmodel.py:50-80)
    want = {"conv_offset_aff.weight": (24, 8, 3, 3), "conv_offset_aff.bias": (24,), "aff_scale_const": (1,), "w": (1, 1, 3, 3), "b": (1,),
            "w_conf": (1, 1, 1, 1)}
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want
    assert float(m.aff_scale_const.detach()) == 4.0 and m.aff_scale_const.requires_grad                 # affinity_gamma * num (:62-64)
    assert not m.w.requires_grad and not m.b.requires_grad and not m.w_conf.requires_grad
    assert float(m.conv_offset_aff.weight.abs().max()) == 0.0                                    # zero init (:55-56)
    assert float(NLSPN(_args(affinity="TC"), 8, 1, 3, 3).aff_scale_const.detach()) == 8.0 and float(NLSPN(_args(affinity="AS"), 8, 1, 3, 5).aff_scale_const.detach()) == 1.0
    with pytest.raises(AssertionError):
        NLSPN(_args(), 8, 2, 3, 3)                       # only ch_f == 1 (:30)
    with pytest.raises(AssertionError):
        NLSPN(_args(), 8, 1, 3, 4)                       # odd kernels only (:35)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP device"):
            m(torch.zeros(1, 1, 8, 8), torch.zeros(1, 8, 8, 8), torch.zeros(1, 1, 8, 8))
        from diffusiondepth_amd import dcn
        with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
            dcn.modulated_deform_conv_forward(torch.zeros(1, 1, 4, 4), torch.ones(1, 1, 3, 3), torch.zeros(1), torch.zeros(1, 18, 4, 4),
                                              torch.ones(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 64)


@pytest.mark.parametrize("name", NLSPN_CASES)
def test_nlspn_mirror_tensor_formulation_matches_reference_goldens(golden, name, monkeypatch):
    """Host logic of diffusiondepth_amd.nlspn.NLSPN's autograd path (channel regrouping, --legacy shift, tanh / gamma, confidence
    weighting, normalisation, preserve_input blend) with the DCNv2 operator replaced by the oracle: no GPU involved."""
    import diffusiondepth_amd.nlspn as N

    class OracleFn:
        @staticmethod
        def apply(inp, off, mask, w, b, stride, pad, dil, groups, dg, step):
            y = O.mdcn_forward(inp.detach().numpy(), w.detach().numpy(), b.detach().numpy(), off.detach().numpy(), mask.detach().numpy(),
                               pad=(pad, pad))
            return torch.from_numpy(y.astype(np.float32))

    monkeypatch.setattr(N, "ModulatedDeformConvFunction", OracleFn)
    g = golden("nlspn_" + name)
    B, H, W, ch_g, k_f, T, cp, pi, lg = [int(v) for v in g["meta"]]
    m = N.NLSPN(_args(prop_time=T, affinity=str(g["affinity"]), conf_prop=bool(cp), preserve_input=bool(pi), legacy=bool(lg)), ch_g, 1, 3, k_f)
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        m.conv_offset_aff.weight.copy_(t("conv_weight"))
        m.conv_offset_aff.bias.copy_(t("conv_bias"))
        off, aff = m._get_offset_affinity(t("guidance"), t("confidence") if cp else None)
        keep = (t("feat_fix") > 0).any(1, keepdim=True) if pi else None
        feat = t("feat_init")
        for _ in range(T):
            if keep is not None:
                feat = torch.where(keep, t("feat_fix"), feat)
            feat = m._propagate_once(feat, off, aff)
    assert float((off - t("offset")).abs().max()) < 1e-6 and float((aff - t("aff")).abs().max()) < 1e-6
    assert float((feat - t("y")).abs().max()) < TOL * np.abs(g["y"]).max()
