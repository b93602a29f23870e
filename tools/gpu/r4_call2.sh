#!/bin/bash
# Round 4, call 2: the refined-f16 mode with block-scaled int16 hand-overs (EK_F16Q forms) and the conv1 forms; lane skew; regression check of the
# bf16 / f16 lines after the epilogue refactor; grid barrier probe 2 (bounded).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest (f16r subset, defaults: f16r_wide=2, f16r_c1=1)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "f16r or full_size or big_tile or native" > gpurun_out/pytest_f16r.log 2>&1; echo "pytest rc=$?"; tail -n 8 gpurun_out/pytest_f16r.log
cp gpurun_out/parity_report.jsonl gpurun_out/parity_report_call2.jsonl 2>/dev/null
X="--no-train-extra --no-nlspn-extra --no-head-extra"
Y="$X --no-abs-extra --no-latency-b1 --no-streams-extra --no-parity-gate"
run() { name=$1; shift; timeout 300 python bench.py --steps 10 --warmup 2 $Y "$@" > gpurun_out/bench_$name.log 2>&1; python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/bench_{n}.log") if l.startswith("{")][-1]); c = d["cpu_baseline"]; r = d["roofline"]
    print(f"{n:22s} {d['value']:8.1f} maps/s  step {d['ms_per_step']:.3f} ms  layers {r['per_layer_avg_us']}  near {c['gpu_vs_cpu_depth_rmse']:.3e}  far {c['far_range']['gpu_vs_cpu_depth_rmse']:.3e}")
except Exception as e:
    print(n, "FAILED", e); print(open(f"gpurun_out/bench_{n}.log").read()[-1500:])
PY
}
run q15_c1w    --set f16r_wide=2 --set f16r_c1=1
run q15_c1none --set f16r_wide=2 --set f16r_c1=0
run q15_c1full --set f16r_wide=2 --set f16r_c1=2
run fp32_c1w   --set f16r_wide=1 --set f16r_c1=1
run narrow_c1w --set f16r_wide=0 --set f16r_c1=1
run q15_skew90   --set lane_skew_us=90
run q15_skew180  --set lane_skew_us=180
run q15_skew270  --set lane_skew_us=270
run bf16 --precision bf16
run bf16_skew180 --precision bf16 --set lane_skew_us=180
run f16  --precision f16
echo "== bench f16r default line (full)"; timeout 600 python bench.py --steps 20 --warmup 3 $X > gpurun_out/bench_f16r.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_f16r.log | cut -c1-600
echo "== grid barrier probe 2"
timeout 100 build_variants/grid_sync_probe2 > gpurun_out/grid_sync_probe2.txt 2>&1; echo "rc=$?"; cat gpurun_out/grid_sync_probe2.txt
