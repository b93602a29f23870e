#!/bin/bash
# Round 4, call 23: rocprofv3 --kernel-trace --stats of the bf16 mode on one stream (round 3's like-for-like figure: conv3 150.8 us = 0.335, conv2 143.3 us = 0.352)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_bf16" -o bench --output-format csv -- python "$R/bench.py" --steps 5 --warmup 1 --streams 1 --precision bf16 --batch 4 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra --no-abs-extra --no-parity-gate > "$R/gpurun_out/rocprof_bf16.log" 2>&1); echo "rocprof rc=$?"
KS=$(find gpurun_out/prof_bf16 -name "*kernel_stats.csv" | head -1); head -n 10 "$KS" | cut -c1-170
find gpurun_out/prof_bf16 -name "*kernel_trace.csv" -delete
tail -n 1 gpurun_out/rocprof_bf16.log | cut -c1-300
