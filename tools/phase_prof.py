#!/usr/bin/env python3
"""Per-workgroup phase timeline of one convolution layer (where does a workgroup's lifetime go?).

Builds a variant of the library with -DDD_PHASE_PROF=1 into build_variants/ (every workgroup's thread 0 stores the 100 MHz wall clock at
kernel entry / GroupNorm table done / first patch + weights in LDS / main loop done / stores issued / exit, plus HW_ID and XCC_ID), runs
the eager loop and summarises: phase durations, workgroup lifetime, concurrency per CU, launch span.

    python tools/phase_prof.py build                      (here: hipcc cross-compiles)
    python tools/phase_prof.py run [layers] [B] [prec] [res|swin]    (on the GPU box; layers e.g. 2,9,1,4)
"""
import os, subprocess, sys
os.environ.setdefault("DDEPTH_STREAMS", "1")      # kernel-level measurements: one stream (the binding defaults to two concurrent lanes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
VARIANT = os.path.join(ROOT, "build_variants", "libddepth_hip_prof.so")


def build():
    from diffusiondepth_amd import build as b
    os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
    hipcc = b.find_hipcc()
    objs = []
    for src in b.SOURCES:
        obj = os.path.join(ROOT, "build_variants", "prof_" + os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + [f for f in b.FLAGS if f != "-shared"] + ["-DDD_PHASE_PROF=1"] + os.environ.get("DDEPTH_CFLAGS", "").split() + ["-x", "hip", "-c", os.path.join(b.CSRC, src), "-o", obj]
        if src not in ("dd_api.cpp", "dd_api_weights.cpp", "dd_api_plans.cpp", "dd_api_train.cpp", "dd_igemm2.hip") and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(os.path.join(b.CSRC, src)):
            objs.append(obj); continue
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", VARIANT] + objs)
    print(VARIANT)


def run(layers, B, prec, variant="res"):
    os.environ["DDEPTH_LIBRARY"] = VARIANT
    import numpy as np, torch
    import diffusiondepth_amd as dda
    from diffusiondepth_amd import synth
    h, w, T = 176, 608, 3
    be = dda.HipDenoiser(variant=variant); be.load_state_dict(synth.make_state_dict(7240 if variant == "res" else 7245, variant)); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    be.set_option("graph", 0)
    inp = synth.make_inputs(7240, B, h, w, None if variant == "res" else ((352 + 3) // 4, (1216 + 3) // 4))      # Swin: the stride-4 condition map
    x, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
    buf = torch.zeros(8 * 16384 + 4 * 8 * 16384, dtype=torch.int64, device="cuda")
    for _ in range(2):
        be.denoise(x, cond, T, prec)
    for layer in layers:
        buf.zero_()
        be.set_option("phase_prof_layer", layer); be.set_option("phase_prof_buffer", buf.data_ptr())
        be.denoise(x, cond, T, prec)
        torch.cuda.synchronize()
        be.set_option("phase_prof_buffer", 0)
        raw = buf.cpu().numpy()
        a = raw.reshape(-1, 8)
        n = int((a[:, 0] > 10 ** 9).sum())          # workgroup records (wall-clock values) are contiguous from 0; the per-wave cycle sums follow
        if n == 0:
            print(f"\n== layer {layer}: no launch of this layer in the loop"); continue
        a = a[:n]
        t = (a[:, :6] - a[:, 0].min()) / 100.0            # us since the first workgroup started
        ph = np.diff(t, axis=1)
        names = ["entry->table", "table->patch0+W0 in LDS", "main loop", "epilogue stores", "stats+exit"]
        print(f"\n== layer {layer}, B={B}, {prec}: {n} workgroups, launch span {t[:, 5].max():.1f} us (last launch of the eager loop)")
        for i, nm in enumerate(names):
            v = ph[:, i]
            print(f"  {nm:26s} median {np.median(v):7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}  mean {v.mean():7.2f} us")
        life = t[:, 5] - t[:, 0]
        print(f"  {'workgroup lifetime':26s} median {np.median(life):7.2f}  p10 {np.percentile(life, 10):7.2f}  p90 {np.percentile(life, 90):7.2f}  mean {life.mean():7.2f} us")
        hw, xcc = a[:, 6], a[:, 7] & 0xF
        cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)      # cu_id, sh_id, se_id, xcc
        ncu = len(np.unique(cu))
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        print(f"  distinct CUs {ncu}; workgroups per CU min {per_cu.min()} max {per_cu.max()}; sum of lifetimes / (CUs x span) = "
              f"{life.sum() / (ncu * t[:, 5].max()):.2f} resident workgroups per CU on average")
        # concurrency over time: how many workgroups are in their main loop at once
        ev = np.concatenate([np.stack([t[:, 2], np.ones(n)], 1), np.stack([t[:, 3], -np.ones(n)], 1)])
        ev = ev[np.argsort(ev[:, 0])]
        conc = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0], append=ev[-1, 0])
        print(f"  time-average number of workgroups inside the main loop: {float((conc * dt).sum() / t[:, 5].max()):.1f} (of {2 * ncu} slots at 2 per CU)")
        starts = np.sort(t[:, 0])
        print(f"  workgroup start times: first wave of {int((starts < 1.0).sum())} within 1 us; 50 % started by {starts[n // 2]:.1f} us; last start {starts[-1]:.1f} us")
        waves = 8 if layer in (1, 4) else 4
        lw = raw[n * 8: n * 8 + n * waves * 4].reshape(-1, 4).astype(np.float64)
        lw = lw[lw.sum(1) > 0]
        if len(lw):
            tot = lw.sum(1)
            print(f"  per-wave shader-clock cycles inside the main loop (mean over {len(lw)} waves): MFMA blocks {lw[:, 0].mean():.0f}, prologue transform {lw[:, 1].mean():.0f}, "
                  f"waits + barrier {lw[:, 2].mean():.0f}, DMA / raw-load issue {lw[:, 3].mean():.0f}; total {tot.mean():.0f} (= {tot.mean() / np.median(ph[:, 2]) / 1e3:.2f} GHz x main-loop median)")
        np.save(os.path.join(ROOT, "gpurun_out", f"phase_prof_layer{layer}_b{B}_{prec}.npy"), a)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        layers = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2,9,1,4").split(",")]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        run(layers, int(sys.argv[3]) if len(sys.argv) > 3 else 4, sys.argv[4] if len(sys.argv) > 4 else "bf16", sys.argv[5] if len(sys.argv) > 5 else "res")
