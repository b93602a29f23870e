#!/bin/bash
cd "$(dirname "$0")"; mkdir -p gpurun_out; export TMPDIR=/tmp
for b in 2 3 4 6 8; do
  timeout 300 python bench.py --steps 8 --warmup 2 --precision bf16 --batch $b --no-cpu-baseline > gpurun_out/bench_b.log 2>&1
  python - "$b" <<'PY'
import json,sys
d=json.loads([x for x in open("gpurun_out/bench_b.log") if x.startswith("{")][-1]); r=d["roofline"]
print("B="+sys.argv[1], d["value"], "maps/s loop_ms", r["loop_ms_graph"], "per-map loop ms", round(r["loop_ms_graph"]/int(sys.argv[1]),3), "layers_us", r["per_layer_avg_us"])
PY
done
