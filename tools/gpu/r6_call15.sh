#!/bin/bash
# Round 6, call 15: the whole GPU suite on today's tree (Swin f16r single call, facade test, gradient guard, tile rule), smoke, the default bench line
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r06_call15_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -n 12 gpurun_out/r06_call15_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_call15_smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 3 gpurun_out/r06_call15_smoke.txt
timeout 900 python bench.py > gpurun_out/r06_call15_bench_default.json 2> gpurun_out/r06_call15_bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r06_call15_bench_default.json | cut -c1-1500
cp gpurun_out/parity_report.jsonl gpurun_out/r06_call15_parity_report.jsonl 2>/dev/null
