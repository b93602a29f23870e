#!/bin/bash
# Round 4, call 11: lane count of the shipped step in the refined f16 mode (B=4 and B=8), two passes each on one box
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra --no-abs-extra"
for rep in 1 2; do
  for B in 4 8; do
    for S in 1 2 3 4; do
      [ $S -gt $B ] && continue
      timeout 300 python bench.py --steps 10 --warmup 3 --batch $B --streams $S $X > gpurun_out/bench_lanes_b${B}_s${S}_$rep.log 2>&1
      python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_lanes_b${B}_s${S}_$rep.log").read().strip().splitlines()[-1])
    print("[lanes] B=$B S=$S rep=$rep", d["value"], "maps/s", d["ms_per_step"], "ms")
except Exception as e:
    print("[lanes] B=$B S=$S rep=$rep failed", e)
PY
    done
  done
done
