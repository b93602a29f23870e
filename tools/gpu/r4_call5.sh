#!/bin/bash
# Round 4, call 5: batch sweep (suggested_batch), NYU, the Swin denoiser's f16 mode at KITTI's depth range, barrier probe diagnostics
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--no-train-extra --no-nlspn-extra --no-head-extra --no-abs-extra --no-latency-b1 --no-streams-extra --no-parity-gate"
run() { name=$1; shift; timeout 400 python bench.py --steps 10 --warmup 2 $X "$@" > gpurun_out/bench_$name.log 2>&1; python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/bench_{n}.log") if l.startswith("{")][-1]); c = d.get("cpu_baseline") or {}; r = d["roofline"]
    far = c.get("far_range", {})
    print(f"{n:22s} {d['value']:8.1f} maps/s  step {d['ms_per_step']:.3f} ms  step_frac {r['step_frac_of_peak']}  layers {r['per_layer_avg_us']}  near {c.get('gpu_vs_cpu_depth_rmse', float('nan')):.3e}  far {far.get('gpu_vs_cpu_depth_rmse', float('nan')):.3e} (max depth {far.get('depth_range_m', [0, 0])[1]})")
except Exception as e:
    print(n, "FAILED", e); print(open(f"gpurun_out/bench_{n}.log").read()[-1500:])
PY
}
run kitti_b6  --batch 6 --no-cpu-baseline
run kitti_b8  --batch 8 --no-cpu-baseline
run kitti_b16 --batch 16 --no-cpu-baseline --steps 5
run nyu_b4    --size nyu
run nyu_b16   --size nyu --batch 16 --no-cpu-baseline
run nyu_b28   --size nyu --batch 28 --no-cpu-baseline
run nyu_b28_bf16 --size nyu --batch 28 --no-cpu-baseline --precision bf16
run swin_f16  --variant swin --precision f16 --steps 5
run swin_bf16 --variant swin --precision bf16 --steps 5
echo "== grid barrier probe 2"
timeout 100 build_variants/grid_sync_probe2 > gpurun_out/grid_sync_probe2c.txt 2>&1; echo "rc=$?"; head -8 gpurun_out/grid_sync_probe2c.txt
