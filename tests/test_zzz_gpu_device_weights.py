"""GPU (`-m gpu`), last in collection order on purpose: the device route of the denoiser parameters (dd_set_weight_device -> pack kernels of
dd_misc.hip; include/ddepth.h).  It is what keeps a training iteration's parameter refresh in HBM after optimizer.step(); it has been
validated bit for bit against the host packer under host emulation (tests/test_library_host_emulation.py); this file passed on an
MI355X in round 2 (profiles/history/r02_run10_pytest_gpu.txt), since when the route is the default (DDEPTH_DEVICE_WEIGHTS=1)."""
import numpy as np
import pytest
import torch

import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

pytestmark = pytest.mark.gpu


def _cuda_sd(sd):
    return {k: torch.from_numpy(v).cuda() for k, v in sd.items()}


@pytest.mark.parametrize("variant", ["res", "swin"])
def test_pack_kernels_leave_the_host_packers_bytes_in_hbm(variant):
    sd = synth.make_state_dict(7301, variant)
    host, dev = dda.HipDenoiser(variant=variant), dda.HipDenoiser(variant=variant)
    host.load_state_dict(sd, device_route=False)
    dev.load_state_dict(_cuda_sd(sd), device_route=True)
    assert dev.weights_digest() == host.weights_digest()
    # a second, different set through the device route (buffers reused), against a fresh host load
    upd = {k: (v * 1.25 + 0.01).astype(np.float32) for k, v in sd.items() if k.startswith("model.")}
    dev.load_state_dict(_cuda_sd(upd), device_route=True)
    fresh = dda.HipDenoiser(variant=variant)
    fresh.load_state_dict({**sd, **upd}, device_route=False)
    assert dev.weights_digest() == fresh.weights_digest() != host.weights_digest()


def test_training_refresh_through_the_module_tree_stays_on_the_device(monkeypatch):
    """ScheduledCNNRefine in .train(): forward + backward, an SGD step, forward again -- with DDEPTH_DEVICE_WEIGHTS=1 the refreshed
    parameters reach the kernels without a host copy; the result equals a host-loaded backend holding the updated values, bit for bit."""
    monkeypatch.setenv("DDEPTH_DEVICE_WEIGHTS", "1")
    sd = synth.make_state_dict(7244)
    model = dda.ScheduledCNNRefine(precision="bf16")
    model.load_state_dict({k[len("model."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("model.")})
    model = model.cuda().train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    inp = synth.make_inputs(41, 2, 13, 21)
    x, cond, t = (torch.from_numpy(inp[k]).cuda() for k in ("x_T", "cond", "timesteps"))
    eps = model(x, t, cond, None, None, None)
    eps.square().mean().backward()
    opt.step()
    with torch.no_grad():
        after = model(x, t, cond, None, None, None)                 # HipBound sees the bumped versions: device-route refresh
    ref = dda.HipDenoiser()
    ref.load_state_dict({"model." + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, device_route=False)
    ref.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    assert model.bound.backend.weights_digest() == ref.weights_digest()
    want = ref.denoise_once(x, t, cond, "bf16")
    assert float((after - want).abs().max()) <= 1e-6 * float(want.abs().max())      # same bytes, same kernels (fp64 statistics atomics: order)
    assert not torch.equal(after, eps.detach())                     # and the step did change the network
