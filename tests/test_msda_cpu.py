"""CPU: the multi-scale deformable attention of the HAHI neck (reference src/model/necks/hahi.py:108-118,211-247).
  * the oracle (oracle/msda_oracle.py, "parity unpinned": mmcv-full is un-vendored and absent) evaluated two independent ways -- the operator's
    definition in NumPy fp64 against the grid_sample formulation -- incl. samples outside the maps, on the border cells, one-row / one-column levels;
  * the product's host-side mirror (diffusiondepth_amd/msda.py: module, positional encoding; necks.py: the neck with attention ON) in eager mode
    on CPU tensors against the oracle's module restatement;
  * the reference's dead end reproduced: four inputs (three transformer levels) against num_levels = 4 fail to broadcast, five inputs run;
  * the C ABI of include/ddepth_msda.h: every declared symbol exported and bound with the prototype's arity; argument validation answers before
    any HIP call.
No compute calls into the library here (no GPU in this container)."""
import os
import re

import numpy as np
import pytest
import torch

from diffusiondepth_amd import msda, necks
from oracle import msda_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_case(seed, B, M, D, shapes, Q, P, spread=0.35):
    """Random operands; sampling locations spread beyond [0, 1] so that whole samples and single corners fall outside the maps."""
    r = np.random.RandomState(seed)
    L = len(shapes)
    K = sum(h * w for h, w in shapes)
    value = r.standard_normal((B, K, M, D)).astype(np.float32)
    loc = (r.uniform(-spread, 1 + spread, (B, Q, M, L, P, 2))).astype(np.float32)
    loc[0, 0, 0, 0, 0] = (0.0, 0.0)                      # exactly the map's corner: x = y = -0.5
    loc[0, 0, 0, 0, min(1, P - 1)] = (1.0, 1.0)
    attn = r.uniform(0, 1, (B, Q, M, L * P)).astype(np.float32)
    attn = (attn / attn.sum(-1, keepdims=True)).reshape(B, Q, M, L, P)
    return value, np.asarray(shapes, np.int64), loc, attn


CASES = [(1, 2, 2, 8, [(5, 7), (3, 4)], 6, 3), (2, 1, 8, 64, [(6, 8), (3, 4), (2, 2), (1, 1)], 11, 4), (3, 2, 3, 5, [(1, 9), (7, 1)], 4, 2)]


@pytest.mark.parametrize("seed,B,M,D,shapes,Q,P", CASES)
def test_oracle_definition_equals_grid_sample_formulation(seed, B, M, D, shapes, Q, P):
    value, sh, loc, attn = make_case(seed, B, M, D, shapes, Q, P)
    a = O.ms_deform_attn_core(value, sh, loc, attn)
    b = O.ms_deform_attn_core_grid_sample(value, sh, loc, attn).numpy()
    assert a.shape == (B, Q, M * D)
    assert np.abs(a - b).max() < 1e-12 * max(1.0, np.abs(a).max())
    # and the product's eager form (what CPU tensors run) in fp32
    c = msda.multi_scale_deformable_attn_pytorch(torch.from_numpy(value), torch.from_numpy(sh), torch.from_numpy(loc), torch.from_numpy(attn)).numpy()
    assert np.abs(a - c).max() < 2e-6 * max(1.0, np.abs(a).max())


def test_operator_properties():
    """Linear in the attention weights and in the value; a sample outside (-1, H) x (-1, W) contributes nothing; a location on a cell centre
    returns that cell."""
    value, sh, loc, attn = make_case(7, 1, 2, 4, [(4, 5)], 3, 2)
    base = O.ms_deform_attn_core(value, sh, loc, attn)
    assert np.allclose(O.ms_deform_attn_core(2 * value, sh, loc, attn), 2 * base) and np.allclose(O.ms_deform_attn_core(value, sh, loc, 3 * attn), 3 * base)
    far = loc.copy(); far[..., 0] = 1.3                      # x = 1.3 * 5 - 0.5 = 6 >= W: outside
    assert np.abs(O.ms_deform_attn_core(value, sh, far, attn)).max() == 0.0
    one = np.zeros((1, 1, 2, 1, 1, 2), np.float32); one[..., 0] = (2 + 0.5) / 5; one[..., 1] = (1 + 0.5) / 4
    got = O.ms_deform_attn_core(value, sh, one, np.ones((1, 1, 2, 1, 1), np.float32)).reshape(2, 4)
    assert np.allclose(got, value[0].reshape(4, 5, 2, 4)[1, 2], atol=1e-6)


def _module_and_params(E=32, heads=4, levels=2, points=3, seed=0):
    torch.manual_seed(seed)
    m = msda.MultiScaleDeformableAttention(embed_dims=E, num_heads=heads, num_levels=levels, num_points=points, batch_first=True).eval()
    with torch.no_grad():                                # away from the zero-weight initialisation: every term of the module matters
        for p in m.parameters():
            p.add_(0.1 * torch.randn_like(p))
    return m, {k: v.detach().numpy() for k, v in m.state_dict().items()}


def test_module_forward_eager_vs_oracle():
    E, heads, levels, points = 32, 4, 2, 3
    m, p = _module_and_params(E, heads, levels, points)
    shapes = [(4, 6), (2, 3)]
    K, Q, B = 30, 9, 2
    r = np.random.RandomState(3)
    query, qpos, value = (r.standard_normal(s).astype(np.float32) for s in ((B, Q, E), (B, Q, E), (B, K, E)))
    ref_pts = r.uniform(0, 1, (B, Q, levels, 2)).astype(np.float32)
    sh = torch.as_tensor(shapes)
    starts = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
    with torch.no_grad():
        got = m(torch.from_numpy(query), value=torch.from_numpy(value), query_pos=torch.from_numpy(qpos), reference_points=torch.from_numpy(ref_pts),
                spatial_shapes=sh, level_start_index=starts).numpy()
    want = O.msda_module_forward(p, query, value, None, qpos, ref_pts, shapes, None, heads, levels, points)
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
    # self attention: value = query (WITHOUT the positional term), identity = query
    with torch.no_grad():
        got = m(torch.from_numpy(value), query_pos=torch.from_numpy(np.zeros_like(value)), reference_points=torch.from_numpy(r.uniform(0, 1, (B, K, levels, 2)).astype(np.float32)),
                spatial_shapes=sh, level_start_index=starts)
    assert got.shape == (B, K, E)
    # initialisation of the module (Deformable DETR's): zero offset weights, directions in the bias, uniform attention
    fresh = msda.MultiScaleDeformableAttention(embed_dims=E, num_heads=heads, num_levels=levels, num_points=points)
    assert float(fresh.sampling_offsets.weight.detach().abs().max()) == 0.0 and float(fresh.attention_weights.bias.detach().abs().max()) == 0.0
    b = fresh.sampling_offsets.bias.view(heads, levels, points, 2)
    assert torch.allclose(b[0, 0, :, 0], torch.arange(1, points + 1, dtype=torch.float32)) and torch.allclose(b[:, :, 1], 2 * b[:, :, 0])


def test_sine_positional_encoding_vs_oracle():
    mask = torch.zeros((2, 5, 7), dtype=torch.bool)
    mask[1, :, 5:] = True
    for kw in (dict(num_feats=8), dict(num_feats=6, normalize=True, offset=-0.5)):
        got = msda.SinePositionalEncoding(**kw)(mask).numpy()
        want = O.sine_positional_encoding(mask.numpy(), **kw)
        keep = ~mask.numpy()[:, None]        # (inside a padded column the normalised row coordinate is -0.5 / 1e-6: sin / cos of 3e6, fp32 noise; the neck's masks are all False)
        assert got.shape == (2, 2 * kw["num_feats"], 5, 7) and (np.abs(got - want) * keep).max() < 1e-5
    with pytest.raises(KeyError):
        msda.build_positional_encoding(dict(type="LearnedPositionalEncoding", num_feats=8))


def _neck(n_in, seed=1):
    torch.manual_seed(seed)
    chans = [8] + [16] * (n_in - 1)
    n = necks.HAHIHeteroNeck(chans, chans, 32, positional_encoding=dict(type="SinePositionalEncoding", num_feats=16), cross_att=True, self_att=True, num_points=2)
    n.init_weights()
    with torch.no_grad():
        for m in (n.multi_att, n.self_attn):
            for p in m.parameters():
                p.add_(0.05 * torch.randn_like(p))
        for m in n.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    return n.eval(), chans


def test_neck_with_attention_runs_for_five_inputs_and_fails_for_four_as_the_reference_does():
    n, chans = _neck(5)
    x = [torch.randn(2, chans[0], 12, 16)] + [torch.randn(2, 16, max(1, 12 >> i), max(1, 16 >> i)) for i in range(1, 5)]
    with torch.no_grad():
        outs = n(x)
    assert [tuple(o.shape) for o in outs] == [tuple(t.shape) for t in x]
    # the self attention against the oracle's module restatement, fed with the neck's own intermediate tensors
    with torch.no_grad():
        feats = [c(x[i]) for i, c in enumerate(n.lateral_convs)]
        srcs, poss, shapes = [], [], []
        for i, f in enumerate(feats[1:]):
            b, _, h, w = f.shape
            shapes.append((h, w))
            poss.append(n.trans_positional_encoding(torch.zeros((b, h, w), dtype=torch.bool)).flatten(2).transpose(1, 2) + n.level_embed[i].view(1, 1, -1))
            srcs.append(n.trans_proj[i](f).flatten(2).transpose(1, 2))
        src, pos = torch.cat(srcs, 1), torch.cat(poss, 1)
        sh = torch.as_tensor(shapes)
        starts = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
        ref = n.get_reference_points(shapes, torch.ones(2, 4, 2), "cpu")
        got = n.self_attn(src, query_pos=pos, reference_points=ref, spatial_shapes=sh, level_start_index=starts).numpy()
    want = O.msda_module_forward({k: v.numpy() for k, v in n.self_attn.state_dict().items()}, src.numpy(), None, None, pos.numpy(),
                                 O.reference_points_of_levels(shapes, np.ones((2, 4, 2))), shapes, None, 8, 4, 2)
    assert np.abs(got - want).max() < 5e-5 * np.abs(want).max()
    assert np.abs(ref.numpy() - O.reference_points_of_levels(shapes, np.ones((2, 4, 2)))).max() < 1e-6
    # the configuration the DiffusionDepth heads would feed it: 3 transformer levels against num_levels = 4
    n4, chans4 = _neck(4)
    with pytest.raises(RuntimeError, match="must match the size"):
        n4(x[:4])
    # ... which is why they switch attention off; that path is untouched (and keeps the attention parameters for checkpoints)
    off = necks.HAHIHeteroNeck(chans4, chans4, 32, positional_encoding=dict(type="SinePositionalEncoding", num_feats=16), cross_att=False, self_att=False, num_points=2).eval()
    assert [tuple(o.shape) for o in off(x[:4])] == [tuple(t.shape) for t in x[:4]]
    assert {"multi_att.sampling_offsets.weight", "self_attn.output_proj.bias", "reference_points.weight", "level_embed"} <= set(off.state_dict())


def test_library_exports_every_msda_symbol_with_the_prototypes_arity():
    import diffusiondepth_amd as dda
    hdr = open(os.path.join(ROOT, "include", "ddepth_msda.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(dd_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(msda.ABI_SYMBOLS), declared ^ set(msda.ABI_SYMBOLS)
    lib = msda._lib()
    dda.load_library()
    text = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name, params in re.findall(r"(?:int|const char\*)\s+(dd_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        assert hasattr(lib, name), name
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        fn = getattr(lib, name)
        assert fn.argtypes is not None and len(fn.argtypes) == n, (name, n, len(fn.argtypes or ()))
    assert lib.dd_msda_last_error() == b""
    assert lib.dd_msda_forward(*([None] * 6), 1, 4, 1, 8, 1, 2, 1, 64, None) == 1 and b"null tensor pointer" in lib.dd_msda_last_error()
    assert lib.dd_msda_backward(*([None] * 9), 1, 4, 1, 8, 1, 2, 1, 64, None) == 1
    # the operator itself has no CPU path: CPU tensors never reach it (they take the module's eager form)
    v = torch.zeros(1, 4, 1, 8)
    with pytest.raises(RuntimeError, match="HIP device"):
        msda.MultiScaleDeformableAttnFunction.apply(v, torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 2, 1, 1, 1, 2), torch.zeros(1, 2, 1, 1, 1), 64)
