#!/bin/bash
# One gpurun call: diagnostics -> parity tests (separate processes per group so a device fault in one
# precision mode cannot take the others down) -> smoke -> bench -> rocprofv3 kernel stats.
# Everything is logged under gpurun_out/ (merged back into the build container).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== diag";   timeout 600 python tests/gpu_diag.py > gpurun_out/diag.log 2>&1; echo "diag rc=$?"
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 25 gpurun_out/pytest_gpu.log
echo "== smoke";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -n 5 gpurun_out/smoke.log
echo "== bench";  timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_bf16.log 2>&1; echo "bench rc=$?"
tail -n 3 gpurun_out/bench_bf16.log

timeout 300 python bench.py --steps 10 --warmup 2 --batch 16 --no-cpu-baseline > gpurun_out/bench_bf16_b16.log 2>&1; tail -n 1 gpurun_out/bench_bf16_b16.log
timeout 300 python bench.py --steps 5 --warmup 2 --variant swin --no-cpu-baseline > gpurun_out/bench_bf16_swin.log 2>&1; tail -n 1 gpurun_out/bench_bf16_swin.log
for b in 1 4; do timeout 300 python tools/head_timing.py $b bf16 2>&1 | grep "B=" ; done | tee gpurun_out/head_timing.log
timeout 300 python bench.py --steps 10 --warmup 2 --precision f16 --no-cpu-baseline > gpurun_out/bench_f16.log 2>&1; tail -n 1 gpurun_out/bench_f16.log
timeout 300 python bench.py --steps 5 --warmup 1 --precision fp32 --no-cpu-baseline > gpurun_out/bench_fp32.log 2>&1; tail -n 1 gpurun_out/bench_fp32.log
echo "== rocprof"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_bf16" -o bench --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-train-extra --no-latency-b1 > "$OLDPWD/gpurun_out/rocprof_bf16.log" 2>&1); echo "rocprof rc=$?"
find gpurun_out/prof_bf16 -name "*stats*" | head; 
for f in $(find gpurun_out/prof_bf16 -name "*kernel_stats.csv" | head -1); do head -n 20 "$f"; done
# keep the merge-back small: drop the raw traces, keep the stats
find gpurun_out/prof_bf16 -name "*kernel_trace.csv" -size +20M -delete
ls -la gpurun_out | head -30
