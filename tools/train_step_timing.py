#!/usr/bin/env python3
"""Training-step timing at KITTI latent size: the T-step loop forward + backward through the HIP library (autograd
Functions of diffusiondepth_amd.modules) vs the same module tree run as plain PyTorch-ROCm ops with torch autograd.
    python tools/train_step_timing.py [batch] [T] [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
h, w = 176, 608
sd = synth.make_state_dict(7240)
model = dda.ScheduledCNNRefine(precision=prec)
model.load_state_dict({k[len("model."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("model.")})
model = model.cuda().train()
sched = dda.DDIMScheduler()
pipe = dda.CNNDDIMPipiline(model, sched)
inp = synth.make_inputs(1, B, h, w)
x_T = torch.from_numpy(inp["x_T"]).cuda()
cond = torch.from_numpy(inp["cond"]).cuda().requires_grad_(True)
wts = torch.randn_like(x_T)

def hip_step():
    model.zero_grad(set_to_none=True); cond.grad = None
    x0, = pipe(batch_size=B, device=cond.device, dtype=torch.float32, shape=(16, h, w), input_args=(cond, None, None, None),
               num_inference_steps=T, return_dict=False, x_T=x_T)
    (wts * x0).sum().backward()

def torch_denoiser(x, t, feat):                      # the same parameters through stock torch ops (reference ...res.py:324-344)
    f = feat + model.time_embedding(t)[..., None, None] + model.noise_embedding(x)
    return model.pred(f)

def torch_step(autocast):
    model.zero_grad(set_to_none=True); cond.grad = None
    sched.set_timesteps(T)
    x = x_T
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        for t in sched.timesteps:
            eps = torch_denoiser(x, t.cuda(), cond).float()
            x = sched.step(eps, t, x, eta=0.0, use_clipped_model_output=True)["prev_sample"]
    (wts * x).sum().backward()

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

flops = 626688.0 * B * h * w * T
t_hip = timeit(hip_step)
print(f"B={B} T={T} {prec}: HIP loop forward+backward {t_hip:.1f} ms ({3 * flops / t_hip / 1e9:.0f} TFLOP/s counting fwd + dgrad + wgrad; nothing is recomputed)")
try:
    t_t32 = timeit(lambda: torch_step(False), 2)
    t_t16 = timeit(lambda: torch_step(True), 2)
    print(f"           PyTorch-ROCm autograd, same modules: fp32 {t_t32:.1f} ms | autocast bf16 {t_t16:.1f} ms   -> HIP is {t_t32 / t_hip:.1f}x / {t_t16 / t_hip:.1f}x faster")
except Exception as e:
    print("           PyTorch-ROCm baseline failed:", repr(e)[:200])

# ---- the whole iteration: forward + backward + optimizer.step() + the parameter refresh the next forward triggers (HipBound) ----
# The figures above leave the weights untouched between iterations, so the refresh never runs; a real training loop pays it every step.
opt = torch.optim.SGD(model.parameters(), lr=1e-6)

def hip_iteration():
    hip_step()
    opt.step()

for route in ("0", "1"):
    os.environ["DDEPTH_DEVICE_WEIGHTS"] = route
    try:
        t_it = timeit(hip_iteration)
        print(f"           iteration with optimizer.step(), parameter refresh by the {'device' if route == '1' else 'host'} route: {t_it:.1f} ms "
              f"(+{t_it - t_hip:.1f} ms over forward+backward alone)")
    except Exception as e:
        print(f"           iteration with the {'device' if route == '1' else 'host'} route failed:", repr(e)[:200])
