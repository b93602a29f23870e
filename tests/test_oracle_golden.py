"""CPU: pin the NumPy oracle (oracle/ddim_oracle.py) to the known-answer vectors minted from the
reference's own classes (tests/golden/make_golden.py).  Tolerances are fp32-vs-fp64 round-off of the
reference's own fp32 run, scaled to the magnitude of the quantity checked."""
import numpy as np
import pytest

from diffusiondepth_amd import synth
from oracle import ddim_oracle as O


def _sd(c):
    return synth.make_state_dict(c["wseed"], c.get("variant", "res"), c.get("decoder_gain", 0.05),
                                 c.get("decoder_log_scale", 0.0))


def test_schedule_tables_bit_exact(golden):
    g = golden("sched")
    s = O.DDIMScheduleOracle()
    assert np.array_equal(s.betas, g["betas"])
    assert np.array_equal(s.alphas_cumprod, g["alphas_cumprod"])
    # values quoted in BASELINE.md section 4
    assert abs(float(s.alphas_cumprod[0]) - 0.9998999834) < 1e-9
    assert abs(float(s.alphas_cumprod[999]) - 4.0358e-05) < 1e-9


def test_torch_linspace_restatement_matches_torch():
    import torch
    for (a, b, n) in [(1e-4, 0.02, 1000), (1e-4, 0.02, 100), (0.0, 1.0, 7)]:
        assert np.array_equal(O._torch_linspace_f32(a, b, n), torch.linspace(a, b, n, dtype=torch.float32).numpy())


@pytest.mark.parametrize("T", [5, 20, 50])
def test_timesteps_and_step(golden, cases, T):
    g = golden("sched")
    c = cases["sched"]
    s = O.DDIMScheduleOracle()
    ts = s.set_timesteps(T)
    assert np.array_equal(ts, g[f"timesteps_T{T}"])
    rs = np.random.RandomState(c["seed"])
    x = rs.standard_normal(c["shape"]).astype(np.float32)
    eps = np.abs(rs.standard_normal(c["shape"])).astype(np.float32)
    for i, t in enumerate(ts):
        lit = s.step(eps.astype(np.float64), t, x.astype(np.float64))
        c1, c2 = s.coeffs(t)
        closed = c1 * x + c2 * eps
        # literal fp64 == closed form; reference fp32 step() within fp32 round-off (amplified by 1/sqrt(abar_t))
        assert np.abs(lit - closed).max() < 1e-12
        assert np.abs(lit - g[f"step_T{T}"][i]).max() < 2e-5


def test_coeff_table_values_from_baseline():
    s = O.DDIMScheduleOracle()
    s.set_timesteps(20)
    want = {950: (1.5963930, -0.5964435), 900: (1.5564218, -0.5565388), 500: (1.2718236, -0.2863844),
            50: (1.0153209, -0.1660005), 0: (1.0000500, -0.0100013)}
    for t, (c1, c2) in want.items():
        g1, g2 = s.coeffs(t)
        assert abs(g1 - c1) < 2e-6 and abs(g2 - c2) < 2e-6
    prod = np.prod([s.coeffs(t)[0] for t in s.timesteps])
    assert abs(prod - 97.109) < 5e-3


def test_add_noise(golden, cases):
    c = cases["sched"]
    rs = np.random.RandomState(c["seed"])
    rs.standard_normal(c["shape"]); rs.standard_normal(c["shape"])
    B = len(c["add_noise_t"])
    x0 = rs.standard_normal((B,) + tuple(c["shape"][1:])).astype(np.float32)
    nz = rs.standard_normal((B,) + tuple(c["shape"][1:])).astype(np.float32)
    got = O.DDIMScheduleOracle().add_noise(x0, nz, c["add_noise_t"])
    assert np.abs(got - golden("sched")["add_noise"]).max() < 1e-6


@pytest.mark.parametrize("name", ["denoise_res", "denoise_swin"])
def test_denoiser_single_call(golden, cases, name):
    c, g = cases[name], golden(name)
    sd = _sd(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], c.get("cond_hw"))
    eps, mid = O.denoiser_forward(sd, inp["x_T"], c["t"], inp["cond"], c["variant"], return_intermediates=True)
    assert np.abs(eps - g["eps_scalar_t"]).max() < 2e-5
    assert np.abs(mid["ne"][:1, :8] - g["ne_sample0_ch0_8"]).max() < 2e-5
    eps_b = O.denoiser_forward(sd, inp["x_T"], inp["timesteps"], inp["cond"], c["variant"])
    assert np.abs(eps_b - g["eps_batch_t"]).max() < 2e-5
    assert eps.min() >= 0.0          # final GN+ReLU: predicted noise is non-negative


@pytest.mark.parametrize("name", ["loop_res", "loop_res_far", "loop_swin"])
def test_ddim_loop_and_decode(golden, cases, name):
    c, g = cases[name], golden(name)
    sd = _sd(c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], c.get("cond_hw"))
    for T in c["T"]:
        x0 = O.ddim_loop(sd, inp["x_T"], inp["cond"], T, c["variant"])
        ref = g[f"x0_T{T}"]
        scale = np.abs(ref).max()
        # the reference's fp32 run carries ~1e-6 relative round-off on |x_0| ~ 1e2
        assert np.abs(x0 - ref).max() < 3e-6 * scale, (np.abs(x0 - ref).max(), scale)
        d = O.decode(sd, x0, dtype=np.float64)
        dref = g[f"depth_T{T}"]
        assert np.abs(d - dref).max() < 1e-3, np.abs(d - dref).max()      # the north-star tolerance
        assert (np.abs(d - dref) / np.maximum(dref, 1e-2)).max() < 5e-5


def test_codec(golden, cases):
    c, g = cases["codec"], golden("codec")
    sd = _sd(c)
    for i, (B, H, W) in enumerate(c["sizes"]):
        gt = synth.make_gt_depth(c["iseed"] + i, B, H, W)
        lat = O.encode(sd, gt)
        assert lat.shape == g[f"latent_{i}"].shape == (B, 16) + synth.latent_hw(H, W)
        assert np.abs(lat - g[f"latent_{i}"]).max() < 2e-5   # fp32 round-off of pre-tanh values ~|80 m * w|
        h, w = synth.latent_hw(H, W)
        z = np.random.RandomState(c["iseed"] + 100 + i).standard_normal((B, 16, h, w)).astype(np.float32) * c["latent_scale"]
        d = O.decode(sd, z)
        dref = g[f"depth_{i}"]
        assert d.shape == dref.shape == (B, 1, 2 * h, 2 * w)
        assert (np.abs(d - dref) / np.maximum(np.abs(dref), 1e-2)).max() < 2e-5


# ---- the torch-CPU port (oracle/torch_cpu_port.py: cpu_baseline + full-size checker) is pinned too ----
def test_torch_port_loop_and_codec(golden, cases):
    import torch
    from oracle import torch_cpu_port as P
    c, g = cases["loop_res"], golden("loop_res")
    sd = P.to_torch_sd(_sd(c))
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    assert torch.equal(P.make_alphas_cumprod(), torch.from_numpy(golden("sched")["alphas_cumprod"]))
    for T in c["T"]:
        x0 = P.ddim_loop(sd, inp["x_T"], inp["cond"], T).numpy()
        ref = g[f"x0_T{T}"]
        assert np.abs(x0 - ref).max() < 3e-6 * np.abs(ref).max()
        assert np.abs(P.decode(sd, torch.from_numpy(ref)).numpy() - g[f"depth_T{T}"]).max() < 1e-5
    cc, gc = cases["codec"], golden("codec")
    sdc = P.to_torch_sd(_sd(cc))
    gt = synth.make_gt_depth(cc["iseed"], *cc["sizes"][0])
    assert np.abs(P.encode(sdc, gt).numpy() - gc["latent_0"]).max() < 2e-6


def test_torch_port_denoiser_batch_t(golden, cases):
    import torch
    from oracle import torch_cpu_port as P
    c, g = cases["denoise_res"], golden("denoise_res")
    sd = P.to_torch_sd(_sd(c))
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    with torch.no_grad():
        e = P.denoiser(sd, torch.from_numpy(inp["x_T"]), torch.from_numpy(inp["timesteps"]), torch.from_numpy(inp["cond"]))
    assert np.abs(e.numpy() - g["eps_batch_t"]).max() < 1e-5


def test_torch_port_swin_loop(golden, cases):
    from oracle import torch_cpu_port as P
    c, g = cases["loop_swin"], golden("loop_swin")
    sd = P.to_torch_sd(_sd(c))
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], c["cond_hw"])
    x0 = P.ddim_loop(sd, inp["x_T"], inp["cond"], 20, variant="swin").numpy()
    ref = g["x0_T20"]
    assert np.abs(x0 - ref).max() < 3e-6 * np.abs(ref).max()


def test_fpn_aggregate_odd_pyramid(golden, cases):
    """Condition FPN restatement vs the reference head's own modules on an odd-sized pyramid
    (29x39 <- 15x20 <- 8x10 <- 4x5: both adaptive_avg_pool2d size fixes are exercised)."""
    c, g = cases["fpn_odd"], golden("fpn_odd")
    fsd = synth.make_fpn_state_dict(c["fseed"])
    fp = synth.make_backbone_features(c["iseed"], c["B"], c["H"], c["W"])
    x = O.fpn_aggregate(fsd, fp)
    assert list(x.shape) == list(g["shape"])
    scale = float(np.abs(g["cond_ch0_8"]).max())
    assert np.abs(x[:, :8] - g["cond_ch0_8"]).max() <= 2e-6 * scale
    ref_sum = g["cond_chan_sum"]
    assert np.abs(x.sum(axis=(0, 2, 3)) - ref_sum).max() <= 2e-6 * float(np.abs(ref_sum).max())


def test_fpn_aggregate_even_pyramid(golden, cases):
    """Same restatement vs the condition map the reference head computed inside the head_res golden run."""
    c, g = cases["head_res"], golden("head_res")
    fsd = synth.make_fpn_state_dict(c["fseed"])
    fp = synth.make_backbone_features(c["iseed"], c["B"], c["H"], c["W"])
    x = O.fpn_aggregate(fsd, fp)
    scale = float(np.abs(g["cond_ch0_4"]).max())
    assert np.abs(x[:, :4] - g["cond_ch0_4"]).max() <= 2e-6 * scale
    assert abs(float(x.sum()) - float(g["cond_sum"][0])) <= 2e-6 * abs(float(g["cond_sum"][0]))


def test_adaptive_avg_pool_matches_torch():
    import torch
    rs = np.random.RandomState(5)
    for (H, W, oh, ow) in [(30, 40, 29, 39), (16, 20, 15, 20), (8, 10, 8, 10), (7, 9, 3, 4), (5, 5, 7, 6)]:
        x = rs.standard_normal((2, 3, H, W))
        ref = torch.nn.functional.adaptive_avg_pool2d(torch.from_numpy(x), (oh, ow)).numpy()
        assert np.abs(O.adaptive_avg_pool2d(x, oh, ow) - ref).max() < 1e-12


def test_torch_port_denoiser_vjp_matches_reference_autograd(golden, cases):
    """The differentiable torch port (backward oracle) vs autograd of the reference's own ScheduledCNNRefine."""
    import torch
    from oracle import torch_cpu_port as P
    c, g = cases["denoise_bwd_res"], golden("denoise_bwd_res")
    sd = P.to_torch_sd(synth.make_state_dict(c["wseed"], "res"))
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    ge = np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32)
    eps, gx, gc, grads = P.denoiser_vjp(sd, inp["x_T"], torch.from_numpy(inp["timesteps"]), inp["cond"], ge)
    assert np.abs(eps.numpy() - g["eps"]).max() <= 1e-5
    assert np.abs(gx.numpy() - g["grad_x"]).max() <= 2e-5 * np.abs(g["grad_x"]).max()
    assert np.abs(gc.numpy()[:, :8] - g["grad_cond_ch0_8"]).max() <= 2e-5 * np.abs(g["grad_cond_ch0_8"]).max()
    for k in list(g):
        if not k.startswith("grad.model.") or k.endswith((".rows", ".sums")):
            continue
        name = k[len("grad."):]
        if name.endswith(".stride7"):
            got = grads[name[:-len(".stride7")]].numpy().reshape(-1)[::7]
        elif name == "model.time_embedding.weight":
            got = grads[name].numpy()[g[k + ".rows"]]
        else:
            got = grads[name].numpy()
        assert np.abs(got - g[k]).max() <= 5e-5 * max(1e-6, np.abs(g[k]).max()), name


def test_torch_port_swin_denoiser_vjp_matches_reference_autograd(golden, cases):
    import re
    import torch
    from oracle import torch_cpu_port as P
    c, g = cases["denoise_bwd_swin"], golden("denoise_bwd_swin")
    sd = P.to_torch_sd(synth.make_state_dict(c["wseed"], "swin"))
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], tuple(c["cond_hw"]))
    ge = np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32)
    eps, gx, gc, grads = P.denoiser_vjp(sd, inp["x_T"], torch.from_numpy(inp["timesteps"]), inp["cond"], ge, variant="swin")
    rel = lambda a, b: float(np.abs(np.asarray(a) - b).max() / max(1e-12, np.abs(b).max()))
    assert rel(gx.numpy(), g["grad_x"]) < 5e-5 and rel(gc.numpy()[:, :8], g["grad_cond_ch0_8"]) < 5e-5
    for k in list(g):
        if not k.startswith("grad.model.") or k.endswith((".rows", ".sums")):
            continue
        name = k[len("grad."):]
        m = re.match(r"(.*)\.stride(\d+)$", name)
        if m:
            got = grads[m.group(1)].numpy().reshape(-1)[::int(m.group(2))]
        elif name == "model.time_embedding.weight":
            got = grads[name].numpy()[g[k + ".rows"]]
        else:
            got = grads[name].numpy()
        assert rel(got, g[k]) < 1e-4, name


def test_torch_port_loop_vjp_matches_reference_autograd(golden, cases):
    """Differentiable loop of the torch port (loop-backward oracle) vs autograd through the reference's CNNDDIMPipiline."""
    from oracle import torch_cpu_port as P
    c, g = cases["loop_bwd_res"], golden("loop_bwd_res")
    sd = P.to_torch_sd(synth.make_state_dict(c["wseed"], "res"))
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    ge = np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32)
    x0, gx, gc, grads = P.ddim_loop_vjp(sd, inp["x_T"], inp["cond"], ge, T=c["T"])
    rel = lambda a, b: float(np.abs(np.asarray(a) - b).max() / max(1e-12, np.abs(b).max()))
    assert rel(x0.numpy(), g["x0"]) < 1e-5
    assert rel(gx.numpy(), g["grad_xT"]) < 1e-4
    assert rel(gc.numpy()[:, :8], g["grad_cond_ch0_8"]) < 1e-4
    for k in list(g):
        if not k.startswith("grad.model.") or k.endswith(".rows"):
            continue
        name = k[len("grad."):]
        if name.endswith(".stride7"):
            got = grads[name[:-len(".stride7")]].numpy().reshape(-1)[::7]
        elif name == "model.time_embedding.weight":
            got = grads[name].numpy()[g[k + ".rows"]]
        else:
            got = grads[name].numpy()
        assert rel(got, g[k]) < 2e-4, name
