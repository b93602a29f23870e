#!/bin/bash
# Round 4, call 10: the once-per-image pyramid (condition FPN, HAHI neck) of the split / refined f16 modes on the split-f16 kernels (option
# "cond_split") against the fp32-operand kernels it ran on before: parity tests, then the head forward A/B in one process per head.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fpn.py tests/test_zz_gpu_heads.py -x -q 2>&1 | tail -5
timeout 600 python - <<'PY'
import time, torch, bench
import diffusiondepth_amd as dda
dev = torch.device("cuda", 0)
orig = dda.HipDenoiser.condition
for variant in ("res", "swin"):
    for split in (1, 0, 1, 0):
        def cond(self, fp, precision="fp32", export=True, neck=False, _s=split):
            self.set_option("cond_split", _s)
            return orig(self, fp, precision, export, neck)
        dda.HipDenoiser.condition = cond
        r = bench.head_extra(dev, 4, 352, 1216, "f16r", 20, variant)
        print(f"[head {variant} f16r cond_split={split}] inference_only_ms {r['inference_only_ms']} loss_on_device_ms {r['loss_noise_on_device_ms']} maps/s {r['inference_only_maps_per_s']}", flush=True)
dda.HipDenoiser.condition = orig
# the pyramid alone, per call (B=4, KITTI): fp32 / bf16 / f16x3 with and without the switch
from diffusiondepth_amd import synth
for variant, chans, hw in (("res", (64, 128, 256, 512), (352, 1216)), ("swin", (192, 384, 768, 1536), (176, 608))):
    sd = synth.make_state_dict(7240, variant); sd.update(synth.make_fpn_state_dict(7241, in_channels=chans))
    if variant == "swin": sd.update(synth.make_hahi_state_dict(7242, chans))
    be = dda.HipDenoiser(variant=variant); be.load_state_dict(sd); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    fp = [torch.from_numpy(f).to(dev) for f in synth.make_backbone_features(1, 4, hw[0], hw[1], in_channels=chans)]
    for neck in ((False, True) if variant == "swin" else (False,)):
        for prec, split in (("fp32", 1), ("bf16", 1), ("f16x3", 1), ("f16x3", 0)):
            be.set_option("cond_split", split)
            for _ in range(2): be.condition(fp, prec, export=False, neck=neck)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): be.condition(fp, prec, export=False, neck=neck)
            torch.cuda.synchronize()
            print(f"[pyramid {variant} neck={int(neck)} {prec} cond_split={split}] {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call (B=4)", flush=True)
    be.close()
PY
