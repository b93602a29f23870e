#!/bin/bash
# Round 4, call 1: the refined-f16 mode (DD_PREC_F16R) on the GPU for the first time -- parity tests that touch it, then its cost: the bench line in
# f16r (default, with the far-range gate, bf16 / f16x3 side modes), its option variants, and f16 / bf16 beside it; rocprofv3 kernel stats of the f16r line.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest (f16r subset)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "f16r or full_size or big_tile or native" > gpurun_out/pytest_f16r.log 2>&1; echo "pytest rc=$?"; tail -n 8 gpurun_out/pytest_f16r.log
cp gpurun_out/parity_report.jsonl gpurun_out/parity_report_call1.jsonl 2>/dev/null
X="--no-train-extra --no-nlspn-extra --no-head-extra"
echo "== bench f16r (default line)"; timeout 600 python bench.py --steps 20 --warmup 3 $X > gpurun_out/bench_f16r.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_f16r.log | cut -c1-3000
Y="$X --no-abs-extra --no-latency-b1 --no-streams-extra --no-parity-gate"
echo "== bench f16r narrow";  timeout 300 python bench.py --steps 10 --warmup 2 $Y --set f16r_wide=0 > gpurun_out/bench_f16r_narrow.log 2>&1; tail -n 1 gpurun_out/bench_f16r_narrow.log | cut -c1-1800
echo "== bench f16r wide+p4"; timeout 300 python bench.py --steps 10 --warmup 2 $Y --set f16r_p4=1 > gpurun_out/bench_f16r_p4.log 2>&1; tail -n 1 gpurun_out/bench_f16r_p4.log | cut -c1-1800
echo "== bench f16";          timeout 300 python bench.py --steps 10 --warmup 2 $Y --precision f16 > gpurun_out/bench_f16.log 2>&1; tail -n 1 gpurun_out/bench_f16.log | cut -c1-1800
echo "== bench bf16";         timeout 300 python bench.py --steps 10 --warmup 2 $Y --precision bf16 > gpurun_out/bench_bf16.log 2>&1; tail -n 1 gpurun_out/bench_bf16.log | cut -c1-1800
echo "== rocprof f16r (one stream)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_f16r" -o bench --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra > "$OLDPWD/gpurun_out/rocprof_f16r.log" 2>&1); echo "rocprof rc=$?"
for f in $(find gpurun_out/prof_f16r -name "*kernel_stats.csv" | head -1); do head -n 14 "$f" | cut -c1-230; done
find gpurun_out/prof_f16r -name "*kernel_trace.csv" -delete
echo "== grid barrier probe 2 (sc1 data path, two-level barrier)"
timeout 120 build_variants/grid_sync_probe2 > gpurun_out/grid_sync_probe2.txt 2>&1; echo "rc=$?"; cat gpurun_out/grid_sync_probe2.txt
