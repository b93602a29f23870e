"""GPU (`-m gpu`): the multi-scale deformable attention of the HAHI neck through the C ABI of include/ddepth_msda.h (csrc/dd_msda.hip) --
  (1) the operator, forward and backward, against the fp64 oracle (oracle/msda_oracle.py: definition / autograd through its grid_sample
      formulation) on ragged cases (samples and single corners outside the maps, one-row / one-column levels, channel counts that are not a
      power of two) and at the neck's own geometry (8 heads x 64 channels, 4 levels, 8 points);
  (2) the module (diffusiondepth_amd.msda.MultiScaleDeformableAttention) on the device against its own eager form on the CPU, values and
      gradients of every parameter;
  (3) the whole neck with attention ON (five inputs) on the device against the CPU run, and the four-input dead end of the reference.
Tolerances: fp32 against fp64, sums of <= L * P * 4 products per output: 2e-6 of max forward, 1e-5 backward (atomics: order of summation).
The oracle is "parity unpinned" (mmcv-full is un-vendored and absent): see its header."""
import numpy as np
import pytest
import torch

from diffusiondepth_amd import msda, necks
from oracle import msda_oracle as O
from test_msda_cpu import CASES, _neck, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def U():
    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a HIP device: the product has no CPU fallback")
    import gpu_util
    return gpu_util


def _run_op(U, value, sh, loc, attn, go):
    starts = torch.cat((torch.zeros(1, dtype=torch.int64), torch.from_numpy(sh).prod(1).cumsum(0)[:-1])).cuda()
    tv, tl, ta = (U.cu(a).requires_grad_(True) for a in (value, loc, attn))
    out = msda.MultiScaleDeformableAttnFunction.apply(tv, torch.from_numpy(sh).cuda(), starts, tl, ta, 64)
    (out * U.cu(go)).sum().backward()
    return out.detach().cpu().numpy(), tv.grad.cpu().numpy(), tl.grad.cpu().numpy(), ta.grad.cpu().numpy()


@pytest.mark.parametrize("seed,B,M,D,shapes,Q,P", CASES + [(4, 1, 2, 96, [(3, 3)], 3, 2), (5, 2, 8, 64, [(22, 76), (11, 38), (6, 19), (3, 10)], 1500, 8)])
def test_operator_forward_backward_vs_oracle(U, seed, B, M, D, shapes, Q, P):
    value, sh, loc, attn = make_case(seed, B, M, D, shapes, Q, P)
    go = np.random.RandomState(seed + 10).standard_normal((B, Q, M * D)).astype(np.float32)
    out, gv, gl, ga = _run_op(U, value, sh, loc, attn, go)
    tv, tl, ta = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (value, loc, attn))
    ref = O.ms_deform_attn_core_grid_sample(tv, sh, tl, ta)
    (ref * torch.from_numpy(go).double()).sum().backward()
    want = ref.detach().numpy()
    if Q <= 16:
        assert np.abs(O.ms_deform_attn_core(value, sh, loc, attn) - want).max() < 1e-12 * max(1.0, np.abs(want).max())      # both oracle forms
    errs = {"out": U.maxabs(out, want) / max(1.0, np.abs(want).max())}
    for got, r, name in ((gv, tv.grad, "grad_value"), (gl, tl.grad, "grad_sampling_loc"), (ga, ta.grad, "grad_attn_weight")):
        errs[name] = U.maxabs(got, r.numpy()) / max(1.0, float(r.abs().max()))
    U.record("msda_op", B=B, M=M, D=D, L=len(shapes), Q=Q, P=P, **errs)
    assert errs["out"] < 2e-6 and max(errs["grad_value"], errs["grad_sampling_loc"], errs["grad_attn_weight"]) < 1e-5, errs


def test_operator_argument_errors_and_skipped_outputs(U):
    value, sh, loc, attn = make_case(1, 2, 2, 8, [(5, 7), (3, 4)], 6, 3)
    shc, starts = torch.from_numpy(sh).cuda(), torch.tensor([0, 35]).cuda()
    with pytest.raises(RuntimeError, match="inconsistent shapes"):
        msda.MultiScaleDeformableAttnFunction.apply(U.cu(value), shc, starts, U.cu(loc[:, :, :1]), U.cu(attn), 64)
    with pytest.raises(RuntimeError, match="float32"):
        msda.MultiScaleDeformableAttnFunction.apply(U.cu(value).double(), shc, starts, U.cu(loc), U.cu(attn), 64)
    tv = U.cu(value)                                       # no gradient wanted for the value: its buffer is never allocated
    tl = U.cu(loc).requires_grad_(True)
    msda.MultiScaleDeformableAttnFunction.apply(tv, shc, starts, tl, U.cu(attn), 64).sum().backward()
    assert tv.grad is None and tl.grad is not None and bool(torch.isfinite(tl.grad).all())


def test_module_on_device_vs_its_eager_form_on_cpu(U):
    """Values and the gradients of every parameter / input: the device path (rocBLAS linears + dd_msda_forward / _backward) against the same module
    on CPU tensors (linears + grid_sample under autograd)."""
    torch.manual_seed(0)
    E, heads, levels, points = 512, 8, 4, 8                 # the neck's module (hahi.py:109-118)
    m = msda.MultiScaleDeformableAttention(embed_dims=E, num_heads=heads, num_levels=levels, num_points=points, batch_first=True).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    shapes = [(12, 20), (6, 10), (3, 5), (2, 3)]
    K = sum(h * w for h, w in shapes)
    B, Q = 2, 77
    r = np.random.RandomState(5)
    query, qpos, value = (r.standard_normal(s).astype(np.float32) for s in ((B, Q, E), (B, Q, E), (B, K, E)))
    ref_pts = r.uniform(0, 1, (B, Q, levels, 2)).astype(np.float32)
    gout = r.standard_normal((B, Q, E)).astype(np.float32)
    sh = torch.as_tensor(shapes)
    starts = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))

    def run(dev):
        mm = m.to(dev)
        for p in mm.parameters():
            p.grad = None
        q, v = (torch.from_numpy(a).to(dev).requires_grad_(True) for a in (query, value))
        out = mm(q, value=v, query_pos=torch.from_numpy(qpos).to(dev), reference_points=torch.from_numpy(ref_pts).to(dev), spatial_shapes=sh.to(dev),
                 level_start_index=starts.to(dev))
        (out * torch.from_numpy(gout).to(dev)).sum().backward()
        res = {"out": out.detach().cpu().numpy(), "grad_query": q.grad.cpu().numpy(), "grad_value": v.grad.cpu().numpy()}
        res.update({"grad." + k: p.grad.cpu().numpy() for k, p in mm.named_parameters()})
        return res

    cpu, dev = run("cpu"), run("cuda")
    m.to("cpu")
    errs = {k: U.maxabs(dev[k], cpu[k]) / max(1e-12, np.abs(cpu[k]).max()) for k in cpu}
    U.record("msda_module", **errs)
    assert max(errs.values()) < 2e-4, errs                  # fp32 GEMMs of K = 512 in two libraries; the operator itself is at 1e-6 (test above)


def test_neck_with_attention_on_device_vs_cpu(U):
    n, chans = _neck(5)
    torch.manual_seed(3)
    x = [torch.randn(2, chans[0], 24, 40)] + [torch.randn(2, 16, max(1, 24 >> i), max(1, 40 >> i)) for i in range(1, 5)]
    with torch.no_grad():
        want = [o.numpy() for o in n(x)]
        got = [o.cpu().numpy() for o in n.cuda()([t.cuda() for t in x])]
    errs = {f"out{i}": U.maxabs(g, w) / max(1e-12, np.abs(w).max()) for i, (g, w) in enumerate(zip(got, want))}
    U.record("msda_neck", **errs)
    assert max(errs.values()) < 1e-4, errs
    n4, _ = _neck(4)
    with pytest.raises(RuntimeError, match="must match the size"):      # the reference's configuration: three transformer levels against num_levels = 4
        n4.cuda()([t.cuda() for t in x[:4]])


def test_operator_rate_at_the_neck_geometry(U):
    """Timing record (no assertion beyond sanity): the Swin-L neck at KITTI size would query the stride-4 map (88 x 304) against four levels; the
    operator is a gather bound by the L2 / HBM path, its algorithmic traffic = value once + locations + weights + output."""
    B, M, D, P = 1, 8, 64, 8
    shapes = [(44, 152), (22, 76), (11, 38), (6, 19)]
    Q = 88 * 304
    value, sh, loc, attn = make_case(9, B, M, D, shapes, 64, P)
    r = np.random.RandomState(1)
    loc = r.uniform(0, 1, (B, Q, M, len(shapes), P, 2)).astype(np.float32)
    attn = np.full((B, Q, M, len(shapes), P), 1.0 / (len(shapes) * P), np.float32)
    starts = torch.cat((torch.zeros(1, dtype=torch.int64), torch.from_numpy(sh).prod(1).cumsum(0)[:-1])).cuda()
    tv, tl, ta, shc = U.cu(value), U.cu(loc), U.cu(attn), torch.from_numpy(sh).cuda()
    f = lambda: msda.MultiScaleDeformableAttnFunction.apply(tv, shc, starts, tl, ta, 64)
    out = f()
    assert bool(torch.isfinite(out).all()) and abs(float(out.mean()) - float(tv.mean())) < 0.05
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    algo = (value.nbytes + loc.nbytes + attn.nbytes + out.numel() * 4)
    gathered = Q * M * len(shapes) * P * 4 * D * 4
    # backward (all three gradients): the same gathers plus the scatter of grad_value (fp32 atomics) and the channel sums
    tv.requires_grad_(True); tl.requires_grad_(True); ta.requires_grad_(True)
    go = torch.ones_like(out)
    o2 = f()
    o2.backward(go, retain_graph=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        o2.backward(go, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    us_b = e0.elapsed_time(e1) / 5 * 1e3
    U.record("msda_rate", us_per_call=us, algorithmic_GBps=algo / us / 1e3, gathered_GBps=gathered / us / 1e3, backward_us_per_call=us_b)
