"""The drop-in claim on the REFERENCE's own facade: the reference's unmodified ``Diffusion_DCbase_Model``
(/root/reference/src/model/diffusion_dcbase_model.py:26-224, loaded by tests/golden/ref_import.py -- from the tree in the build container, from
the bytecode oracle/ref_py/build_ref.py stages under oracle/_ref/py on the GPU box) builds its head through ``HEADS.build(depth_head_cfg)`` with
the kwarg set of :77-91 (type, in_channels, inference_steps, num_train_timesteps, depth_feature_dim, the two loss_cfgs, init_cfg=args).  Here that
registry holds, in turn, the reference's head class and ``diffusiondepth_amd.head``'s class of the same name; both models get the same parameters,
the same backbone, the same sample and the same RNG seed, and ``forward(sample)`` -- the call ``main.py`` makes (src/main.py:232, :434) -- must
agree within the north star's 1e-3 abs on the predicted depth.

CPU test: diffusiondepth_amd's heads evaluate non-HIP tensors with their own torch children (BASELINE configs[0]: the plumbing configuration).
GPU test (-m gpu): the same on cuda:0 in fp32, where diffusiondepth_amd's heads run the HIP library (encoder, FPN / HAHI neck, T-step loop,
decoder, the ddim_loss call), for the Res family and the Swin-HAHI family (README.md:215 headline configuration)."""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))

import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="needs the reference tree or its staged bytecode (oracle/_ref/py)")


class _Pyramid(nn.Module):
    """A four-level stand-in backbone with the given widths at strides 4 / 8 / 16 / 32 (the Swin-L / MPViT pyramids: the backbones stay upstream
    PyTorch per the north star and are not part of the path; both sides of a comparison get the SAME instance weights)."""

    def __init__(self, widths):
        super().__init__()
        self.stem = nn.Conv2d(3, widths[0], 4, 4)
        self.down = nn.ModuleList(nn.Conv2d(a, b, 2, 2) for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, x):
        x = torch.relu(self.stem(x))
        out = [x]
        for d in self.down:
            x = torch.relu(d(x))
            out.append(x)
        return out


def _facade():
    from diffusiondepth_amd import model as M
    fac = ref_import.load_reference_facade()
    bb = sys.modules["model.backbone"]
    bb.BACKBONE_FACTORIES.update(M.BACKBONES)
    bb.BACKBONE_FACTORIES["swin_l_standin"] = lambda: _Pyramid([192, 384, 768, 1536])
    return fac


def _registries(family):
    """(registry of the reference's head classes, registry of diffusiondepth_amd's), each under the reference's registered names"""
    from diffusiondepth_amd import head as H
    ref_reg, our_reg = ref_import.new_registry("reference heads"), ref_import.new_registry("diffusiondepth_amd heads")
    ns = ref_import.load_reference()
    ref_reg.register_module()(ns.DDIMDepthEstimate_Res)
    if family == "swin_hahi":
        import make_golden_hahi as G
        _, _, SwinHAHI, _ = G.load_hahi_reference()
        ref_reg.register_module()(SwinHAHI)
    for name, cls in H.HEADS.items():
        our_reg.register_module(name=name)(cls)
    return ref_reg, our_reg


def _args(head, backbone, T):
    # the attributes Diffusion_DCbase_Model.__init__ reads (src/config.py flags; diffusion_dcbase_model.py:67-81)
    return SimpleNamespace(backbone_module="mmbev_resnet", backbone_name=backbone, head_specify=head, inference_steps=T, num_train_timesteps=1000)


def _sample(B, H, W, dev, seed=5):
    g = torch.Generator().manual_seed(seed)
    rgb = torch.randn(B, 3, H, W, generator=g)
    gt = torch.rand(B, 1, H, W, generator=g) * 9.0 + 0.5
    dep = gt * (torch.rand(B, 1, H, W, generator=g) < 0.05)
    return {k: v.to(dev) for k, v in dict(rgb=rgb, gt=gt, dep=dep, depth_map=dep.clone(), depth_mask=(dep > 0).float()).items()}


def _build_pair(family, head, backbone, T, dev):
    fac = _facade()
    ref_reg, our_reg = _registries(family)
    torch.manual_seed(11)
    fac.HEADS = ref_reg
    ref_model = fac.Diffusion_DCbase_Model(_args(head, backbone, T))
    # the head's parameters: the seeded synthetic weights of the head goldens (tests/golden/cases.json: a decoder gain that keeps the untrained loop's
    # output in a few units -- with torch's default initialisation the 20-step loop amplifies the last bits of two different fp32 convolution
    # implementations to metres); the backbone keeps its (shared) random initialisation with non-trivial BatchNorm statistics
    from diffusiondepth_amd import synth
    chans = {"res": (64, 128, 256, 512), "swin_hahi": (192, 384, 768, 1536)}[family]
    hsd = dict(synth.make_state_dict(7240, "res" if family == "res" else "swin", 0.05, 0.0))
    hsd.update({k: v for k, v in synth.make_fpn_state_dict(7241 if family == "res" else 7243, in_channels=chans).items() if family == "res" or not k.startswith("convup_fp")})
    if family == "swin_hahi":
        hsd.update(synth.make_hahi_state_dict(7242, chans))
    missing, unexpected = ref_model.depth_head.load_state_dict({k: torch.from_numpy(v) for k, v in hsd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(3)
    for mod in ref_model.depth_backbone.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
    fac.HEADS = our_reg
    our_model = fac.Diffusion_DCbase_Model(_args(head, backbone, T))
    assert type(our_model.depth_head).__module__.startswith("diffusiondepth_amd"), "the reference facade built diffusiondepth_amd's head"
    assert type(ref_model.depth_head).__module__.startswith("model.head"), "... and the reference's own head on the other side"
    missing, unexpected = our_model.load_state_dict(ref_model.state_dict(), strict=True)      # same parameter tree: state_dicts load unchanged
    assert not missing and not unexpected
    return ref_model.to(dev).eval(), our_model.to(dev).eval()


def _compare(ref_model, our_model, sample, tol_pred=1e-3):
    outs = []
    for m in (ref_model, our_model):
        torch.manual_seed(1234)
        if sample["rgb"].is_cuda:
            torch.cuda.manual_seed_all(1234)
        with torch.no_grad():
            outs.append(m(sample))
    r, o = outs
    assert set(r) == set(o), "the 13-key output dict of the reference head (…res.py:171-177)"
    assert {k for k in r if r[k] is None} == {k for k in o if o[k] is None}
    d_pred = float((r["pred"] - o["pred"]).abs().max())
    assert torch.isfinite(o["pred"]).all() and d_pred <= tol_pred, d_pred
    assert float((r["pred_init"] - o["pred_init"]).abs().max()) <= 1e-4 * float(r["pred_init"].abs().max())
    assert abs(float(r["ddim_loss"]) - float(o["ddim_loss"])) <= 1e-3 * max(1.0, abs(float(r["ddim_loss"])))
    return d_pred


def test_reference_facade_builds_and_runs_our_res_head_on_cpu():
    """Res family through the reference's facade on CPU tensors (our head: its own torch children; no HIP device here)."""
    ref_model, our_model = _build_pair("res", "DDIMDepthEstimate_Res", "mmbev_res18", T=5, dev=torch.device("cpu"))
    d = _compare(ref_model, our_model, _sample(1, 64, 96, torch.device("cpu")))
    assert d <= 1e-4        # (same fp32 torch operators on both sides)


@pytest.mark.gpu
@pytest.mark.parametrize("family,head,backbone,T,hw", [
    ("res", "DDIMDepthEstimate_Res", "mmbev_res18", 20, (96, 160)),
    ("swin_hahi", "DDIMDepthEstimate_Swin_ADDHAHI", "swin_l_standin", 20, (96, 160)),
])
def test_reference_facade_forward_matches_with_our_heads_on_the_gpu(family, head, backbone, T, hw):
    """forward(sample) of the reference's facade with the HIP heads registered, against the same facade with the reference's heads, cuda:0, fp32 (the
    heads' default profile "reference"): 1e-3 abs on the predicted depth, both head families."""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from gpu_util import record
    dev = torch.device("cuda", 0)
    ref_model, our_model = _build_pair(family, head, backbone, T, dev)
    d = _compare(ref_model, our_model, _sample(2, hw[0], hw[1], dev))
    be = our_model.depth_head._bound
    assert be.backend is not None and be.backend.counter("graph_launches") + be.backend.counter("eager_loops") > 0, "the T-step loop ran in the HIP library"
    record(f"reference_facade_{family}", depth_max_abs=d)
