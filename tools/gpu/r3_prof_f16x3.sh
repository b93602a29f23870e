#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_f16x3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_f16x3" -o bench --output-format csv -- python "$OLDPWD/bench.py" --precision f16x3 --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra > "$OLDPWD/gpurun_out/rocprof_f16x3.log" 2>&1); echo "rocprof rc=$?"
for f in $(find gpurun_out/prof_f16x3 -name "*kernel_stats.csv" | head -1); do head -n 10 "$f" | cut -c1-200; done
find gpurun_out/prof_f16x3 -name "*kernel_trace.csv" -delete
