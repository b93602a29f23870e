#!/bin/bash
# default bench line on the stamped traffic file; rocprofv3 kernel stats of the Swin f16 line (one stream)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_c3_bf16_b4.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_c3_bf16_b4.log | cut -c1-1500
echo "== rocprof swin f16"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_swin" -o bench --output-format csv -- python "$OLDPWD/bench.py" --variant swin --precision f16 --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra > "$OLDPWD/gpurun_out/rocprof_swin.log" 2>&1); echo "rocprof rc=$?"
for f in $(find gpurun_out/prof_swin -name "*kernel_stats.csv" | head -1); do head -n 14 "$f" | cut -c1-220; done
find gpurun_out/prof_swin -name "*kernel_trace.csv" -delete
