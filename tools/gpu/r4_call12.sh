#!/bin/bash
# Round 4, call 12: the refined mode's conv3(cond) reading the caller's NCHW condition tensor in place (option "cond_direct") against the converting route
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "explicit_condition_tensor or full_size or big_tile or ragged" 2>&1 | tail -4
X="--no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-abs-extra"
for rep in 1 2; do
  for d in 1 0; do
    for B in 4 1; do
      timeout 300 python bench.py --steps 10 --warmup 3 --batch $B --set cond_direct=$d $X > gpurun_out/bench_direct${d}_b${B}_$rep.log 2>&1
      python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_direct${d}_b${B}_$rep.log").read().strip().splitlines()[-1])
    print("[direct=$d] B=$B rep=$rep", d["value"], "maps/s", d["ms_per_step"], "ms; one stream:", (d.get("other_stream_count") or {}).get("maps_per_s"), "b1:", (d.get("latency_b1") or {}).get("ms_per_map"))
except Exception as e:
    print("[direct=$d] B=$B rep=$rep failed", e)
PY
    done
  done
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_direct" -o bench --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra --no-abs-extra > "$OLDPWD/gpurun_out/rocprof_direct.log" 2>&1; cd "$OLDPWD"
KS=$(find gpurun_out/prof_direct -name "*kernel_stats.csv" | head -1); head -n 9 "$KS" | cut -c1-170
find gpurun_out/prof_direct -name "*kernel_trace.csv" -delete
