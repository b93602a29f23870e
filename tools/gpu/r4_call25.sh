#!/bin/bash
# Round 4, call 25: priority of the second lane's stream (experiment build: DDEPTH_LANE_PRIO) -- does an asymmetric pair of lanes overlap better than two equals?
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import torch
print("priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
PY
X="--no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra --no-abs-extra"
for rep in 1 2; do
  for prio in "" -1 1; do
    DDEPTH_LANE_PRIO=$prio timeout 300 python bench.py --steps 10 --warmup 3 $X > gpurun_out/bench_prio.log 2>&1
    echo "[prio='$prio' rep=$rep] $(tail -n 1 gpurun_out/bench_prio.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
