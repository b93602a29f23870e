#!/bin/bash
# Round 4, call 4: the whole GPU suite on the refactored library (refined f16 mode, one-buffer conv3, lanes in the plan key), smoke, the default bench line
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 gpurun_out/pytest_gpu.log
echo "== smoke";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 5 gpurun_out/smoke.log
echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_default.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_default.log | cut -c1-2500
