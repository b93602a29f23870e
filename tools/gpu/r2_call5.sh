#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in swin res; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_train_$v" -o tr --output-format csv -- python "$OLDPWD/bench.py" --mode train-dp --variant $v --batch 4 --steps 2 --warmup 1 > "$OLDPWD/gpurun_out/rocprof_train_$v.log" 2>&1); echo "rocprof $v rc=$?"
tail -1 gpurun_out/rocprof_train_$v.log | cut -c1-300
for f in $(find gpurun_out/prof_train_$v -name "*kernel_stats.csv" | head -1); do head -n 26 "$f" | cut -c1-230; done
find gpurun_out/prof_train_$v -name "*kernel_trace.csv" -delete
done
