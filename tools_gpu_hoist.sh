#!/bin/bash
cd "$(dirname "$0")"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "v2" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_gpu.log
for cfg in "1 " "4 " "1 --hoist" "4 --hoist" "16 --hoist"; do set -- $cfg
  timeout 300 python bench.py --steps 8 --warmup 2 --precision bf16 --batch $1 $2 --no-cpu-baseline > gpurun_out/bench_h.log 2>&1
  python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([x for x in open("gpurun_out/bench_h.log") if x.startswith("{")][-1]); r=d["roofline"]
    print("B="+sys.argv[1], sys.argv[2] if len(sys.argv)>2 else "", d["value"], "maps/s loop_ms", r["loop_ms_graph"], "frac", r["loop_frac_of_peak"], "layers_us", r["per_layer_avg_us"])
except Exception as e: print("bench parse failed", e); print(open("gpurun_out/bench_h.log").read()[-1500:])
PY
done
