"""The NLSPN refinement module END TO END on the CPU: diffusiondepth_amd.nlspn.NLSPN and the DCN-extension drop-in diffusiondepth_amd.dcn
(the product's own Python: fused inference path, autograd Function of the DCNv2 operator) on the host-emulated library, against the goldens
minted from the reference's NLSPN class running on the reference's own DCN device code (tests/golden/make_golden_nlspn.py).
Test infrastructure only (tests/hostemu_head.py patches the binding's private guards; the product refuses CPU tensors)."""
import types

import numpy as np
import pytest
import torch

import hostemu_head

TOL = 1e-5


@pytest.fixture(scope="module")
def lib():
    return hostemu_head.load()


@pytest.fixture()
def dcn(lib, monkeypatch):
    return hostemu_head.install_dcn(lib, monkeypatch.setattr)


def rel(a, b):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-12))


def _module(g):
    from diffusiondepth_amd.nlspn import NLSPN
    B, H, W, ch_g, k_f, T, cp, pi, lg = [int(v) for v in g["meta"]]
    a = types.SimpleNamespace(prop_time=T, affinity=str(g["affinity"]), affinity_gamma=0.5, conf_prop=bool(cp), preserve_input=bool(pi), legacy=bool(lg))
    m = NLSPN(a, ch_g, 1, 3, k_f)
    with torch.no_grad():
        m.conv_offset_aff.weight.copy_(torch.from_numpy(g["conv_weight"]))
        m.conv_offset_aff.bias.copy_(torch.from_numpy(g["conv_bias"]))
    return m


@pytest.mark.parametrize("name", ["tgass", "preserve", "k5"])
def test_nlspn_module_fused_path_vs_reference_golden(dcn, lib, golden, name):
    g = golden("nlspn_" + name)
    m = _module(g).eval()
    cp = bool(int(g["meta"][6]))
    n0 = lib.emu_launch_count()
    with torch.no_grad():
        y, y_inter, offset, aff, gamma = m(torch.from_numpy(g["feat_init"]), torch.from_numpy(g["guidance"]),
                                           torch.from_numpy(g["confidence"]) if cp else None, torch.from_numpy(g["feat_fix"]))
    assert lib.emu_launch_count() > n0, "the library did not run"
    scale = np.abs(g["y_inter"]).max()
    assert float(np.abs(offset.numpy() - g["offset"]).max()) < 2e-5 and float(np.abs(aff.numpy() - g["aff"]).max()) < 1e-5
    assert float(np.abs(y.numpy() - g["y"]).max()) / scale < TOL
    assert float(np.abs(torch.stack(y_inter).numpy() - g["y_inter"]).max()) / scale < TOL
    assert len(y_inter) == int(g["meta"][5]) and torch.equal(y_inter[-1], y)
    assert float(gamma) == float(g["aff_const"][0])


def test_nlspn_module_training_path_vs_reference_autograd(dcn, golden):
    """Gradients of sum(y * G) through prop_time iterations of the DCNv2 autograd Function + the confidence sampling, as minted from the
    reference's autograd (tgass golden); 1e-4 relative: chained fp32 atomically-accumulated scatters."""
    g = golden("nlspn_tgass")
    m = _module(g).train()
    feat, guide, conf = (torch.from_numpy(g[k]).requires_grad_(True) for k in ("feat_init", "guidance", "confidence"))
    y, y_inter, offset, aff, _ = m(feat, guide, conf, torch.from_numpy(g["feat_fix"]))
    scale = np.abs(g["y_inter"]).max()
    assert float(np.abs(y.detach().numpy() - g["y"]).max()) / scale < TOL
    (y * torch.from_numpy(g["grad_y"])).sum().backward()
    errs = {"g_feat_init": rel(feat.grad, g["g_feat_init"]), "g_guidance": rel(guide.grad, g["g_guidance"]),
            "g_confidence": rel(conf.grad, g["g_confidence"]), "g_conv_weight": rel(m.conv_offset_aff.weight.grad, g["g_conv_weight"]),
            "g_conv_bias": rel(m.conv_offset_aff.bias.grad, g["g_conv_bias"]), "g_aff_scale_const": rel(m.aff_scale_const.grad, g["g_aff_scale_const"])}
    assert max(errs.values()) < 1e-4, errs
    assert m.w.grad is None and m.b.grad is None
