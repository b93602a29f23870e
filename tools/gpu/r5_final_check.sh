#!/bin/bash
# Round 5, last call: the GPU suite (with the soak tests added after r5_final.sh), smoke and the default line on the committed tree
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_gpu.log
echo "== smoke";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 gpurun_out/smoke.log
echo "== default line (as the driver runs it)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_default.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_driver_default.log | cut -c1-400
