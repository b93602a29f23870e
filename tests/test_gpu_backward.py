"""GPU (`-m gpu`): backward of one epsilon-network evaluation (dd_denoise_once_backward) through
the C ABI against
  (1) the golden gradients minted by autograd of the reference's own ScheduledCNNRefine (denoise_bwd_res.npz),
  (2) torch autograd of the CPU port on other seeded shapes (ragged sizes, batch > 1, repeated timesteps),
  (3) properties: gradients accumulate across calls and clear with zero_grad; linear in grad_eps; loop backward with
      T = 1 == one call.

ReLU ties: the gradient is discontinuous where a GroupNorm+ReLU pre-activation is zero, and with ~10^5..10^6 activations per
call the smallest |pre-activation| is ~1e-6 -- inside the difference between two fp32 implementations, so a max-norm
comparison can fail by a single flipped mask bit (seen: one element at 6e-8, dbeta off by 2 % of its max).  The golden
cases therefore use input seeds whose smallest |pre-activation| is >= 1.5e-5 (tests/golden/cases.json notes).

Tolerances per gradient tensor: fp32 paths (unfused `naive_fp32`, fused `fp32`): max|err| <= 1e-4 x max|reference| (fp32
sums over up to B*h*w*9*Cin products in a different order than torch's).  bf16 / f16 operand modes (activations AND
gradients stored in 16 bits between the kernels): relative L2 error ||got - ref|| / ||ref|| <= 0.2 / 0.06 (measured
0.03-0.09 / 0.03-0.04 on the golden case, up to 0.17 for bf16 on a 9x33 single image).  A 16-bit forward flips the ReLU mask of the ~0.3 % of elements whose pre-activation is within
rounding of zero; under the RANDOM upstream gradient used here every gradient is a sum of random-sign terms, so those few
flips move it by sqrt(flipped / kept) ~ 5 % however fine the arithmetic is (the last layer's dbeta, a plain masked sum of
grad_eps, shows it in isolation).  The fused fp32 mode, same kernels and layouts, matches to 2e-6.  Measured values are
recorded in parity_report.jsonl.
"""
import numpy as np
import pytest
import torch

from diffusiondepth_amd import synth

pytestmark = pytest.mark.gpu

TOL = {"naive_fp32": 1e-4, "fp32": 1e-4, "f16x3": 5e-3, "f16": 6e-2, "bf16": 2e-1}
# "f16x3" (round 6): the split-f16 forward (three f16 MFMAs per product, fp32 tensors: the values the loss sees hold the 1e-3 ABSOLUTE depth bound) with f16
# gradients through the f16 mode's MFMA kernels behind it.  Its ReLU masks and GroupNorm statistics are the exact forward's, so the mask flips that set the
# plain 16-bit modes' 3-9 % (module docstring) are gone: relative L2 <= 5e-3 per gradient tensor (measured 5e-4 .. 8e-4 in the host emulation); with option
# "x3_grad_fp32" the gradients travel as fp32 through the fp32 mode's kernels and the mode is held to the fp32 bound (test_split_f16_backward_with_fp32_gradients)
# K-step SGD trajectory, bf16 mode against fp32 (test_bf16_training_tracks_the_fp32_trajectory_over_sgd_steps): bounds = 2x measured
# (measured, profiles/history/r03_run2_parity_report.jsonl: loss 1.9e-3 / 7.1e-3, step-0 gradients 0.060 / 0.091 worst tensor (the first conv's weight;
#  most tensors 0.01-0.03), accumulated parameter change 0.062 / 0.199 worst tensor, cosine 0.998 / 0.980)
SGD_LOSS_REL = {"res": 4e-3, "swin": 1.5e-2}
SGD_GRAD0_REL = {"res": 0.12, "swin": 0.18}
SGD_DELTA_REL = {"res": 0.125, "swin": 0.40}
SGD_COS_MIN = {"res": 0.996, "swin": 0.96}
PRECS = ["naive_fp32", "fp32", "f16x3", "bf16", "f16"]


@pytest.fixture(scope="module")
def U():
    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a HIP device: the product has no CPU fallback")
    import gpu_util
    gpu_util.KVER = 2
    return gpu_util


def _rel(a, b, prec="fp32"):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if prec in ("bf16", "f16", "f16x3"):
        return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((b ** 2).sum())))
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


@pytest.mark.parametrize("prec", PRECS)
def test_backward_matches_reference_autograd_golden(U, golden, cases, prec):
    c, g = cases["denoise_bwd_res"], golden("denoise_bwd_res")
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    ge = np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32)
    be.zero_grad()
    gx, gc = be.denoise_once_backward(U.cu(inp["x_T"]), U.cu(inp["timesteps"]), U.cu(inp["cond"]), U.cu(ge), prec)
    errs = {"grad_x": _rel(gx.cpu().numpy(), g["grad_x"], prec), "grad_cond": _rel(gc.cpu().numpy()[:, :8], g["grad_cond_ch0_8"], prec),
            "grad_cond_sum": _rel(gc.double().sum(dim=(0, 2, 3)).cpu().numpy(), g["grad_cond_chan_sum"], prec)}
    for k in list(g):
        if not k.startswith("grad.model.") or k.endswith((".rows", ".sums")):
            continue
        name = k[len("grad."):]
        if name.endswith(".stride7"):
            got = be.grad(name[:-len(".stride7")]).cpu().numpy().reshape(-1)[::7]
        elif name == "model.time_embedding.weight":
            full = be.grad(name).cpu().numpy()
            got = full[g[k + ".rows"]]
            assert float(np.abs(full).sum()) == pytest.approx(float(np.abs(got).sum()), rel=1e-6)      # no other row touched
        else:
            got = be.grad(name).cpu().numpy()
        errs[name] = _rel(got, g[k], prec)
    U.record("bwd_golden", prec=prec, **{k.replace("model.", ""): v for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if v > TOL[prec]}
    assert not bad, bad


@pytest.mark.parametrize("prec", ["naive_fp32", "fp32", "f16x3", "bf16"])
@pytest.mark.parametrize("B,h,w,tt", [(1, 9, 33, [500]), (3, 16, 20, [7, 7, 999])])
def test_backward_matches_torch_port_autograd(U, cases, B, h, w, tt, prec):
    from oracle import torch_cpu_port as P
    c = cases["denoise_bwd_res"]
    be = U.backend_for(c)
    sd = P.to_torch_sd(synth.make_state_dict(c["wseed"], "res"))
    inp = synth.make_inputs(70 + B, B, h, w)
    ge = np.random.RandomState(5 + B).standard_normal(inp["x_T"].shape).astype(np.float32)
    t = torch.tensor(tt)
    _, rgx, rgc, rgrads = P.denoiser_vjp(sd, inp["x_T"], t, inp["cond"], ge)
    be.zero_grad()
    gx, gc = be.denoise_once_backward(U.cu(inp["x_T"]), t.cuda(), U.cu(inp["cond"]), U.cu(ge), prec)
    errs = {"grad_x": _rel(gx.cpu().numpy(), rgx.numpy(), prec), "grad_cond": _rel(gc.cpu().numpy(), rgc.numpy(), prec)}
    for name, ref in rgrads.items():
        errs[name] = _rel(be.grad(name).cpu().numpy(), ref.numpy(), prec)
    U.record("bwd_port", prec=prec, B=B, h=h, w=w, **{k.replace("model.", ""): v for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if v > TOL[prec]}
    assert not bad, bad


def test_split_f16_backward_with_fp32_gradients(U, golden, cases):
    """DD_PREC_F16X3 with option "x3_grad_fp32": the split forward differentiated by the fp32 mode's gradient kernels -- the fp32 modes' bound (1e-4 of max)
    against the reference's autograd, for the single call and (5e-3, ReLU ties over chained steps) the loop."""
    c, g = cases["denoise_bwd_res"], golden("denoise_bwd_res")
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    ge = np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32)
    be.set_option("x3_grad_fp32", 1)
    try:
        be.zero_grad()
        gx, gc = be.denoise_once_backward(U.cu(inp["x_T"]), U.cu(inp["timesteps"]), U.cu(inp["cond"]), U.cu(ge), "f16x3")
        errs = {"grad_x": _rel(gx.cpu().numpy(), g["grad_x"]), "grad_cond": _rel(gc.cpu().numpy()[:, :8], g["grad_cond_ch0_8"])}
        errs.update(_golden_param_errs(be, g, "fp32"))
        U.record("bwd_golden_x3_fp32grad", **{k.replace("model.", ""): v for k, v in errs.items()})
        assert not {k: v for k, v in errs.items() if v > 1e-4}, errs
        c2, g2 = cases["loop_bwd_res"], golden("loop_bwd_res")
        be2 = U.backend_for(c2)
        be2.set_option("x3_grad_fp32", 1)
        inp2 = synth.make_inputs(c2["iseed"], c2["B"], c2["h"], c2["w"])
        ge2 = np.random.RandomState(c2["gseed"]).standard_normal(inp2["x_T"].shape).astype(np.float32)
        be2.zero_grad()
        gx2, gc2 = be2.denoise_backward(U.cu(inp2["x_T"]), U.cu(inp2["cond"]), U.cu(ge2), c2["T"], "f16x3", need_grad_xT=True)
        errs2 = {"grad_xT": _rel(gx2.cpu().numpy(), g2["grad_xT"]), "grad_cond": _rel(gc2.cpu().numpy()[:, :8], g2["grad_cond_ch0_8"])}
        errs2.update(_golden_param_errs(be2, g2, "fp32"))
        U.record("loop_bwd_golden_x3_fp32grad", **{k.replace("model.", ""): v for k, v in errs2.items()})
        assert not {k: v for k, v in errs2.items() if v > 5e-3}, errs2
    finally:
        be.set_option("x3_grad_fp32", 0)
        U.backend_for(cases["loop_bwd_res"]).set_option("x3_grad_fp32", 0)


def test_gradients_accumulate_and_clear(U, cases):
    c = cases["denoise_bwd_res"]
    be = U.backend_for(c)
    inp = synth.make_inputs(3, 1, 8, 16)
    ge = np.random.RandomState(1).standard_normal(inp["x_T"].shape).astype(np.float32)
    args = (U.cu(inp["x_T"]), torch.tensor([321]).cuda(), U.cu(inp["cond"]))
    be.zero_grad()
    gx1, gc1 = be.denoise_once_backward(*args, U.cu(ge), "naive_fp32")
    g1 = be.grad("model.pred.0.weight").cpu().numpy()
    gx2, _ = be.denoise_once_backward(*args, U.cu(2.0 * ge), "naive_fp32")                    # linear in grad_eps, accumulates
    g3 = be.grad("model.pred.0.weight").cpu().numpy()
    assert _rel(gx2.cpu().numpy(), 2.0 * gx1.cpu().numpy()) < 1e-5
    assert _rel(g3, 3.0 * g1) < 1e-5
    be.zero_grad()
    assert float(be.grad("model.pred.0.weight").abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        be.denoise_once_backward(inp_cpu := torch.from_numpy(inp["x_T"]), torch.tensor([321]), torch.from_numpy(inp["cond"]), torch.from_numpy(ge))


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_mfma_wgrad_equals_unfused_wgrad(U, cases, prec):
    """dd_wgrad.hip (MFMA, transposed LDS images, fp32 atomics) against the one-thread-per-weight kernel on the SAME 16-bit
    operands: only the summation order differs.  Ragged size (tiles overhang the image), batch 2, all four layer shapes."""
    c = cases["denoise_bwd_res"]
    be = U.backend_for(c)
    inp = synth.make_inputs(17, 2, 19, 45)
    ge = np.random.RandomState(2).standard_normal(inp["x_T"].shape).astype(np.float32)
    args = (U.cu(inp["x_T"]), torch.tensor([100, 900]).cuda(), U.cu(inp["cond"]), U.cu(ge), prec)
    names = ["model.noise_embedding.0.weight", "model.noise_embedding.3.weight", "model.pred.0.weight", "model.pred.3.weight"]
    be.zero_grad(); be.denoise_once_backward(*args)
    fast = {n: be.grad(n).cpu().numpy() for n in names}
    be.set_option("naive_wgrad", 1)
    try:
        be.zero_grad(); be.denoise_once_backward(*args)
        slow = {n: be.grad(n).cpu().numpy() for n in names}
    finally:
        be.set_option("naive_wgrad", 0)
    for n in names:
        assert _rel(fast[n], slow[n]) < 2e-5, n


@pytest.mark.parametrize("prec", ["naive_fp32", "fp32", "f16x3", "bf16"])
def test_loop_backward_matches_reference_autograd_golden(U, golden, cases, prec):
    """dd_denoise_backward (re-run of the loop keeping the T states, then the chain x_{k+1} = c1 x_k + c2 eps(x_k) walked
    backwards with per-step recompute) vs autograd through the reference's CNNDDIMPipiline (loop_bwd_res.npz, T = 5).
    fp32 modes: 5e-3 of max against the golden -- the reference's own fp32 forward and ours differ by ~1e-6 per activation and
    over the T chained steps that is enough to flip one ReLU mask bit of the 256-channel layer (observed: its dbeta / dbias /
    dW at 1.4e-3..2.6e-3, everything else at 1e-4) -- and, independently, the fused fp32 path must equal the unfused path
    to 1e-4 (same states, same masks).  bf16: relative L2 (see module docstring)."""
    c, g = cases["loop_bwd_res"], golden("loop_bwd_res")
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    ge = np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32)
    x0 = be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), c["T"], prec).cpu().numpy()
    be.zero_grad()
    gx, gc = be.denoise_backward(U.cu(inp["x_T"]), U.cu(inp["cond"]), U.cu(ge), c["T"], prec, need_grad_xT=True)
    tol = {"naive_fp32": 5e-3, "fp32": 5e-3, "f16x3": 2e-2, "bf16": 3e-1}[prec]      # (f16x3: relative L2; a ReLU tie flipped over the chained steps shows as ~1e-2 there)
    errs = {"x0": _rel(x0, g["x0"], prec), "grad_xT": _rel(gx.cpu().numpy(), g["grad_xT"], prec),
            "grad_cond": _rel(gc.cpu().numpy()[:, :8], g["grad_cond_ch0_8"], prec),
            "grad_cond_sum": _rel(gc.double().sum(dim=(0, 2, 3)).cpu().numpy(), g["grad_cond_chan_sum"], prec)}
    for k in list(g):
        if not k.startswith("grad.model.") or k.endswith(".rows"):
            continue
        name = k[len("grad."):]
        if name.endswith(".stride7"):
            got = be.grad(name[:-len(".stride7")]).cpu().numpy().reshape(-1)[::7]
        elif name == "model.time_embedding.weight":
            got = be.grad(name).cpu().numpy()[g[k + ".rows"]]
        else:
            got = be.grad(name).cpu().numpy()
        errs[name] = _rel(got, g[k], prec)
    U.record("loop_bwd_golden", prec=prec, **{k.replace("model.", ""): v for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, bad
    if prec == "fp32":
        fused = {n: be.grad(n).cpu().numpy() for n in be.PARAM_SHAPES}
        be.zero_grad()
        gx2, gc2 = be.denoise_backward(U.cu(inp["x_T"]), U.cu(inp["cond"]), U.cu(ge), c["T"], "naive_fp32", need_grad_xT=True)
        assert _rel(gx.cpu().numpy(), gx2.cpu().numpy()) < 1e-4 and _rel(gc.cpu().numpy(), gc2.cpu().numpy()) < 1e-4
        for n in be.PARAM_SHAPES:
            assert _rel(fused[n], be.grad(n).cpu().numpy()) < 1e-4, n


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "bf16"])
def test_loop_backward_on_the_states_the_forward_kept(U, cases, prec):
    """Training path of modules._DenoiseLoopFn: dd_denoise(keep_trajectory) + dd_denoise_backward(use_trajectory = its ticket) skips the
    second forward loop.  Same numbers as the regenerating path in every precision: since round 3 the backward's recompute runs the
    kernels the forward ran (conv3's condition term hoisted in the 16-bit modes: ADVICE r2), so kept and regenerated states / activations
    are the same bytes and both paths differentiate the same function."""
    c = cases["loop_bwd_res"]
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
    ge = U.cu(np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32))
    names = list(be.PARAM_SHAPES)

    def run(keep):
        x0 = be.denoise(x, cond, c["T"], prec, keep_trajectory=keep)
        tk = be.last_trajectory_ticket
        assert (tk > 0) == keep
        be.zero_grad()
        n0 = be.counter("trajectory_reuses")
        gx, gc = be.denoise_backward(x, cond, ge, c["T"], prec, need_grad_xT=True, trajectory_ticket=tk)
        assert be.counter("trajectory_reuses") == n0 + (1 if keep else 0)
        return [x0.cpu().numpy(), gx.cpu().numpy(), gc.cpu().numpy()] + [be.grad(n).cpu().numpy() for n in names]

    regen, kept = run(False), run(True)
    assert np.array_equal(regen[0], kept[0])                  # the forward result does not depend on where the states are written
    # the single call (ddim_loss) likewise: its backward reads the activations the forward left in its (hoisted, in the 16-bit modes) plan
    t = torch.tensor([10, 900][:c["B"]] * (c["B"] // min(2, c["B"])) if c["B"] > 1 else [500]).cuda()[:c["B"]]
    once = {}
    for keep in (False, True):
        eps = be.denoise_once(x, t, cond, prec, keep_trajectory=keep)
        tk = be.last_trajectory_ticket
        be.zero_grad()
        n0 = be.counter("trajectory_reuses")
        gx1, gc1 = be.denoise_once_backward(x, t, cond, ge, prec, trajectory_ticket=tk)
        assert be.counter("trajectory_reuses") == n0 + int(keep)
        once[keep] = [eps.cpu().numpy(), gx1.cpu().numpy(), gc1.cpu().numpy()] + [be.grad(n).cpu().numpy() for n in names]
    assert np.array_equal(once[True][0], once[False][0])
    errs1 = {k: _rel(a, b, prec) for k, a, b in zip(["eps", "grad_x", "grad_cond"] + names, once[True], once[False])}
    U.record("once_bwd_kept_vs_recomputed", prec=prec, **{k.replace("model.", ""): v for k, v in errs1.items()})
    assert not {k: v for k, v in errs1.items() if v > 1e-5}, errs1          # measured 0.0 in both precisions: the same kernels on the same bytes
    errs = {k: _rel(a, b, prec) for k, a, b in zip(["x0", "grad_xT", "grad_cond"] + names, kept, regen)}
    U.record("loop_bwd_kept_vs_regenerated", prec=prec, **{k.replace("model.", ""): v for k, v in errs.items()})
    tol = 1e-5        # (measured 0.0; it used to be "statistics atomics order -> 1e-7 on a state -> at worst one ReLU mask bit" when the recompute ran other kernels)
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, bad


@pytest.mark.parametrize("variant", ["res", "swin"])
def test_bf16_training_tracks_the_fp32_trajectory_over_sgd_steps(U, variant):
    """VERDICT r2 weak #7 / ADVICE r2: the 16-bit training mode against fp32 -- not one gradient under a loose relative-L2 gate, but K = 5
    SGD steps: loss trajectory and the accumulated parameter change.  Both runs go through the library's own training path (forward that
    KEEPS its states / activations + dd_denoise_backward / dd_denoise_once_backward on the ticket, as modules._DenoiseLoopFn /
    _DenoiseOnceFn do); the fp32 run is the reference trajectory: its gradients equal the reference's autograd to 2e-6
    (test_backward_matches_reference_autograd_golden, test_autograd_through_head_modules_matches_torch_port).
    loss = mse(eps(x_q, t), noise) + <g, x_0(T=2 loop)> / numel, plain SGD on every denoiser parameter.
    Also: the step-0 gradients of the bf16 KEPT path against the fp32 ones, per tensor (relative L2 bound) -- the gate ADVICE r2 asked for."""
    import diffusiondepth_amd as dda
    K, T, lr = 5, 2, 0.05
    B, h, w = 2, 24, 40
    cond_hw = (12, 20) if variant == "swin" else None
    inp = synth.make_inputs(41, B, h, w, cond_hw)
    x, cond, noise, t = U.cu(inp["x_T"]), U.cu(inp["cond"]), U.cu(inp["noise"]), U.cu(inp["timesteps"])
    g = U.cu(np.random.RandomState(7).standard_normal(inp["x_T"].shape).astype(np.float32)) / float(inp["x_T"].size)
    sd0 = {k: v for k, v in synth.make_state_dict(7246, variant).items()}

    def run(prec):
        be = dda.HipDenoiser(variant=variant)
        sd = {k: torch.from_numpy(v.copy()).cuda() for k, v in sd0.items()}
        be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
        names = list(be.param_shapes())
        losses, grads0 = [], None
        for k in range(K):
            be.load_state_dict(sd)
            be.zero_grad()
            x0 = be.denoise(x, cond, T, prec, keep_trajectory=True)
            be.denoise_backward(x, cond, g, T, prec, need_grad_xT=False, trajectory_ticket=be.last_trajectory_ticket)
            eps = be.denoise_once(x, t, cond, prec, keep_trajectory=True)
            ge = (2.0 / eps.numel()) * (eps - noise)
            be.denoise_once_backward(x, t, cond, ge.contiguous(), prec, trajectory_ticket=be.last_trajectory_ticket)
            losses.append(float(((eps - noise) ** 2).mean()) + float((g * x0).sum()))
            gr = {n: be.grad(n).clone() for n in names}
            if k == 0:
                grads0 = {n: v.cpu().numpy() for n, v in gr.items()}
            for n in names:
                sd[n] = sd[n] - lr * gr[n].reshape(sd[n].shape)
        out = {n: (sd[n].cpu().numpy() - sd0[n]) for n in names}
        be.close()
        return losses, grads0, out

    l32, g32, d32 = run("fp32")
    l16, g16, d16 = run("bf16")
    assert l32[-1] < l32[0]                                                   # the steps do descend
    loss_rel = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(l16, l32))
    g_rel = {n: _rel(g16[n], g32[n], "bf16") for n in g32}
    d_rel = {n: _rel(d16[n], d32[n], "bf16") for n in d32}
    cos = {n: float((d16[n].astype(np.float64) * d32[n]).sum() / max(1e-30, np.linalg.norm(d16[n].astype(np.float64)) * np.linalg.norm(d32[n].astype(np.float64)))) for n in d32}
    U.record("sgd_trajectory", variant=variant, loss_fp32=l32, loss_bf16=l16, loss_rel_max=loss_rel,
             grad0_rel_max=max(g_rel.values()), grad0_rel_worst=max(g_rel, key=g_rel.get), delta_rel_max=max(d_rel.values()),
             delta_rel_worst=max(d_rel, key=d_rel.get), delta_cos_min=min(cos.values()),
             grad0_rel={k.replace("model.", ""): round(v, 4) for k, v in g_rel.items()})
    assert loss_rel < SGD_LOSS_REL[variant], (loss_rel, l16, l32)
    assert not {n: v for n, v in g_rel.items() if v > SGD_GRAD0_REL[variant]}, g_rel
    assert not {n: v for n, v in d_rel.items() if v > SGD_DELTA_REL[variant]}, d_rel
    assert min(cos.values()) > SGD_COS_MIN[variant], cos


@pytest.mark.parametrize("variant,prec", [("res", "bf16"), ("res", "fp32"), ("swin", "bf16")])
def test_training_pair_as_concurrent_lanes_equals_one_stream(U, cases, variant, prec):
    """dd_set_option("streams", 2) in training: the state-keeping forward and the backward run the batch as two concurrent sub-batches on
    separate HIP streams (per-lane plans, kept activations, gradient sets and weight-gradient workspaces; the sets are summed at the join).
    Per-image results (x_0, grad_xT, grad_cond) are bit-identical to one stream; the parameter gradients are sums over the batch in another
    order.  Three images on two lanes (2 + 1)."""
    import diffusiondepth_amd as dda
    be = dda.HipDenoiser(variant=variant)
    be.load_state_dict(synth.make_state_dict(7240, variant))
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    B, h, w, T = 3, 24, 72, 3
    inp = synth.make_inputs(31, B, h, w, (12, 36)) if variant == "swin" else synth.make_inputs(31, B, h, w)
    x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
    g = U.cu(np.random.RandomState(2).standard_normal(inp["x_T"].shape).astype(np.float32))
    names = list(be.param_shapes())
    res = {}
    try:
        for S in (1, 2):
            be.set_option("streams", S)
            for rep in range(2):                       # second pass: replayed lane graphs, reused buffers
                n0, r0 = be.counter("lane_calls"), be.counter("trajectory_reuses")
                out = be.denoise(x, cond, T, prec, keep_trajectory=True)
                be.zero_grad()
                gx, gc = be.denoise_backward(x, cond, g, T, prec, need_grad_xT=True, trajectory_ticket=be.last_trajectory_ticket)
                assert be.counter("trajectory_reuses") == r0 + 1 and be.counter("lane_calls") == n0 + (2 if S > 1 else 0)
            res[S] = [out, gx, gc] + [be.grad(n) for n in names]
    finally:
        be.set_option("streams", 1)
    for k in range(3):
        assert torch.equal(res[2][k], res[1][k]), ("x0", "grad_xT", "grad_cond")[k]
    bad = {n: _rel(a.cpu().numpy(), b.cpu().numpy()) for n, a, b in zip(names, res[2][3:], res[1][3:])}
    bad = {n: e for n, e in bad.items() if e > 2e-5}
    assert not bad, bad
    be.close()


def test_loop_backward_t1_equals_single_call(U, cases):
    """T = 1: x_0 = c1 x_T + c2 eps(x_T, t = 0)  ->  the loop backward is one denoiser VJP scaled by c2 plus c1 * g."""
    import diffusiondepth_amd as dda
    c = cases["denoise_bwd_res"]
    be = U.backend_for(c)
    inp = synth.make_inputs(8, 1, 8, 40)
    ge = np.random.RandomState(4).standard_normal(inp["x_T"].shape).astype(np.float32)
    x, cond, g = U.cu(inp["x_T"]), U.cu(inp["cond"]), U.cu(ge)
    sched = dda.DDIMScheduler()
    a0 = float(sched.alphas_cumprod[0])
    c1 = (1.0 / a0) ** 0.5
    c2 = 0.0 - ((1.0 - a0) / a0) ** 0.5          # sqrt(1 - abar_prev) - sqrt(abar_prev (1 - abar_t) / abar_t) with abar_prev = 1
    be.zero_grad()
    gx_loop, gc_loop = be.denoise_backward(x, cond, g, 1, "fp32", need_grad_xT=True)
    w_loop = be.grad("model.pred.3.weight").cpu().numpy()
    be.zero_grad()
    gx_once, gc_once = be.denoise_once_backward(x, torch.tensor([0]).cuda(), cond, (c2 * g).contiguous(), "fp32")
    w_once = be.grad("model.pred.3.weight").cpu().numpy()
    assert _rel(gc_loop.cpu().numpy(), gc_once.cpu().numpy()) < 1e-5
    assert _rel(w_loop, w_once) < 1e-5
    assert _rel(gx_loop.cpu().numpy(), (c1 * g + gx_once).cpu().numpy()) < 1e-5


def test_autograd_through_head_modules_matches_torch_port(U, cases):
    """loss.backward() through the drop-in modules (ScheduledCNNRefine + CNNDDIMPipiline in .train() mode, fp32 kernels):
    parameter .grad, cond.grad and the loss value against torch autograd of the CPU port on the same tensors -- the wiring
    the reference's training step relies on (…res.py:124-169, 201-217; src/main.py:232-241)."""
    import diffusiondepth_amd as dda
    from oracle import torch_cpu_port as P
    c = cases["loop_bwd_res"]
    sd = synth.make_state_dict(c["wseed"], "res")
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    T = c["T"]
    # reference side (CPU, torch autograd): loss = sum(w * x_0) + mse(eps(x_q, t), noise)
    wts = torch.from_numpy(np.random.RandomState(3).standard_normal(inp["x_T"].shape).astype(np.float32))
    sdt = P.to_torch_sd(sd)
    params = {k: v.clone().requires_grad_(True) for k, v in sdt.items() if k.startswith("model.")}
    full = dict(sdt); full.update(params)
    cond_r = torch.from_numpy(inp["cond"]).clone().requires_grad_(True)
    acp = P.make_alphas_cumprod()
    cur = torch.from_numpy(inp["x_T"])
    for t in P.timesteps(T):
        cur = P.ddim_step(acp, P.denoiser(full, cur, int(t), cond_r), int(t), cur, 1000 // T)
    tq = torch.from_numpy(inp["timesteps"])
    eps_q = P.denoiser(full, torch.from_numpy(inp["noise"]), tq, cond_r)
    loss_r = (wts * cur).sum() + torch.nn.functional.mse_loss(eps_q, torch.from_numpy(inp["noise"]))
    loss_r.backward()
    # product side
    model = dda.ScheduledCNNRefine(precision="fp32")
    model.load_state_dict({k[len("model."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("model.")})
    model = model.cuda().train()
    sched = dda.DDIMScheduler()
    pipe = dda.CNNDDIMPipiline(model, sched)
    cond = U.cu(inp["cond"]).requires_grad_(True)
    x0, = pipe(batch_size=c["B"], device=cond.device, dtype=torch.float32, shape=(16, c["h"], c["w"]), input_args=(cond, None, None, None),
               num_inference_steps=T, return_dict=False, x_T=U.cu(inp["x_T"]))
    eps = model(U.cu(inp["noise"]), U.cu(inp["timesteps"]), cond, None, None, None)
    loss = (wts.cuda() * x0).sum() + torch.nn.functional.mse_loss(eps, U.cu(inp["noise"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) <= 1e-4 * abs(float(loss_r.detach()))
    errs = {"cond": _rel(cond.grad.cpu().numpy(), cond_r.grad.numpy())}
    named = dict(model.named_parameters())
    for k, v in params.items():
        errs[k] = _rel(named[k[len("model."):]].grad.cpu().numpy(), v.grad.numpy())
    U.record("autograd_modules", **{k.replace("model.", ""): v for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if v > 5e-3}
    assert not bad, bad


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_head_training_step_matches_reference_golden(U, golden, cases, prec):
    """One training-mode step of the drop-in head (BatchNorm on batch statistics in the PyTorch-ROCm FPN / codec, the DDIM
    loop and ddim_loss through the HIP forward AND backward, RNG draws injected) against the same step of the reference head
    run under autograd on CPU (head_train_res.npz): loss, prediction, gradients w.r.t. the backbone features and a sample of
    parameter gradients from every part of the head.  Tolerance 1e-2 of max per tensor (ReLU ties, see module docstring;
    MIOpen vs oneDNN convolutions in the torch-side FPN / codec), loss 1e-4."""
    import diffusiondepth_amd as dda
    c, g = cases["head_train_res"], golden("head_train_res")
    sd = synth.make_state_dict(c["wseed"], "res", c["decoder_gain"], c["decoder_log_scale"])
    sd.update(synth.make_fpn_state_dict(c["fseed"]))
    head = dda.DDIMDepthEstimate_Res(in_channels=[64, 128, 256, 512], inference_steps=c["T"], num_train_timesteps=1000,
                                     depth_feature_dim=16, loss_cfgs=[], precision=prec)
    missing, unexpected = head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected
    head = head.cuda().train()
    B, H, W = c["B"], c["H"], c["W"]
    fp = [U.cu(f).requires_grad_(True) for f in synth.make_backbone_features(c["iseed"], B, H, W)]
    gt = U.cu(synth.make_gt_depth(c["iseed"] + 1, B, H, W))
    h, w = synth.latent_hw(H, W)
    inp = synth.make_inputs(c["iseed"] + 2, B, h, w)
    draws = [U.cu(inp["x_T"]), torch.from_numpy(inp["noise"])]
    real_randn, real_randint = torch.randn, torch.randint
    torch.randn = lambda *a, **k: draws.pop(0)
    torch.randint = lambda *a, **k: U.cu(inp["timesteps"])
    prev = torch.backends.cudnn.allow_tf32
    torch.backends.cudnn.allow_tf32 = False
    try:
        out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
        loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
        loss.backward()
    finally:
        torch.randn, torch.randint = real_randn, real_randint
        torch.backends.cudnn.allow_tf32 = prev
    lv = float(loss.detach())
    assert abs(lv - float(g["loss"][0])) <= 1e-4 * abs(float(g["loss"][0]))
    errs = {"pred": _rel(out["pred"].detach().cpu().numpy(), g["pred"]),
            "grad_fp3": _rel(fp[3].grad.cpu().numpy(), g["grad_fp3"]), "grad_fp0": _rel(fp[0].grad.cpu().numpy()[:, :4], g["grad_fp0_ch0_4"])}
    named = dict(head.named_parameters())
    for k in c["grad_keys"]:
        got = named[k].grad.cpu().numpy()
        if got.size > 5000:
            got = got.reshape(-1)[::c["grad_stride"]]
        errs[k] = _rel(got, g["grad." + k])
    U.record("head_train", prec=prec, loss=lv, **errs)
    bad = {k: v for k, v in errs.items() if v > 1e-2}
    assert not bad, bad


def _golden_param_errs(be, g, prec):
    """{name: error} of every 'grad.model.*' entry of a golden file (strided samples / embedding rows handled)."""
    import re
    errs = {}
    for k in list(g):
        if not k.startswith("grad.model.") or k.endswith((".rows", ".sums")):
            continue
        name = k[len("grad."):]
        m = re.match(r"(.*)\.stride(\d+)$", name)
        if m:
            got = be.grad(m.group(1)).cpu().numpy().reshape(-1)[::int(m.group(2))]
        elif name == "model.time_embedding.weight":
            got = be.grad(name).cpu().numpy()[g[k + ".rows"]]
        else:
            got = be.grad(name).cpu().numpy()
        errs[name] = _rel(got, g[k], prec)
    return errs


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "bf16"])
def test_swin_backward_matches_reference_autograd_golden(U, golden, cases, prec):
    """Swin / MPViT denoiser (UpSample_add fuse: two 256->256 convs without norm, bilinearly upsampled stride-4 condition map):
    gradients vs autograd of the reference's own class (denoise_bwd_swin.npz), incl. the adjoint of the upsample (grad_cond at
    the condition map's own size) and the convA / convB parameter gradients."""
    c, g = cases["denoise_bwd_swin"], golden("denoise_bwd_swin")
    be = U.backend_for(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], tuple(c["cond_hw"]))
    ge = np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32)
    be.zero_grad()
    gx, gc = be.denoise_once_backward(U.cu(inp["x_T"]), U.cu(inp["timesteps"]), U.cu(inp["cond"]), U.cu(ge), prec)
    assert tuple(gc.shape) == tuple(inp["cond"].shape)
    errs = {"grad_x": _rel(gx.cpu().numpy(), g["grad_x"], prec), "grad_cond": _rel(gc.cpu().numpy()[:, :8], g["grad_cond_ch0_8"], prec),
            "grad_cond_sum": _rel(gc.double().sum(dim=(0, 2, 3)).cpu().numpy(), g["grad_cond_chan_sum"], prec)}
    errs.update(_golden_param_errs(be, g, prec))
    U.record("bwd_swin_golden", prec=prec, **{k.replace("model.", ""): v for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if v > TOL[prec]}
    assert not bad, bad


def test_swin_loop_backward_matches_torch_port_autograd(U, cases):
    """Loop backward of the Swin variant (T = 2, fp32 kernels): the chain rule + per-step recompute around the Swin VJP, incl.
    dLoss/dcond accumulated over the steps at the condition map's own size, vs torch autograd of the CPU port."""
    from oracle import torch_cpu_port as P
    c = cases["denoise_bwd_swin"]
    be = U.backend_for(c)
    sd = P.to_torch_sd(synth.make_state_dict(c["wseed"], "swin"))
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], tuple(c["cond_hw"]))
    ge = np.random.RandomState(6).standard_normal(inp["x_T"].shape).astype(np.float32)
    x0, rgx, rgc, rgrads = P.ddim_loop_vjp(sd, inp["x_T"], inp["cond"], ge, T=2, variant="swin")
    for keep in (False, True):       # True: the training path -- states and activations (incl. the convA / convB results) kept by the forward
        tk = 0
        if keep:
            be.denoise(U.cu(inp["x_T"]), U.cu(inp["cond"]), 2, "fp32", keep_trajectory=True)
            tk = be.last_trajectory_ticket
            assert tk > 0
        be.zero_grad()
        n0 = be.counter("trajectory_reuses")
        gx, gc = be.denoise_backward(U.cu(inp["x_T"]), U.cu(inp["cond"]), U.cu(ge), 2, "fp32", need_grad_xT=True, trajectory_ticket=tk)
        assert be.counter("trajectory_reuses") == n0 + int(keep)
        errs = {"grad_xT": _rel(gx.cpu().numpy(), rgx.numpy()), "grad_cond": _rel(gc.cpu().numpy(), rgc.numpy())}
        for name, ref in rgrads.items():
            errs[name] = _rel(be.grad(name).cpu().numpy(), ref.numpy())
        U.record("loop_bwd_swin", kept=keep, **{k.replace("model.", ""): v for k, v in errs.items()})
        bad = {k: v for k, v in errs.items() if v > 5e-3}       # ReLU-tie sensitivity over chained steps, see the Res loop test
        assert not bad, (keep, bad)
