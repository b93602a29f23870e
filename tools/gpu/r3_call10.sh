#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--steps 20 --warmup 3 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra"
for o in "thin_slots=512" "thin_slots=256" "thin_stream=0" "thin_slots=512" "thin_slots=256" "thin_stream=0"; do echo "== $o"; timeout 300 python bench.py $X --set $o 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_us'], d['roofline']['loop_ms_graph'])"; done
