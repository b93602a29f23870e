#!/bin/bash
# Re-take the PMC passes (HBM traffic stamp) on the final sources, then the default bench line (which reads the stamped file) and the GPU suite.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pmc"; bash tools/gpu/pmc.sh > gpurun_out/pmc.log 2>&1; tail -n 3 gpurun_out/pmc.log
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_c3_bf16_b4.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_c3_bf16_b4.log | cut -c1-1500
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_gpu.log
echo "== smoke";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/smoke.log
