#!/bin/bash
# FPN bring-up: FPN parity tests + head golden test + head timing
cd "$(dirname "$0")"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 600 python -m pytest tests/test_gpu_fpn.py -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_fpn.log 2>&1; echo "pytest fpn rc=$?"; tail -n 25 gpurun_out/pytest_fpn.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "v2 and head" > gpurun_out/pytest_head.log 2>&1; echo "pytest head rc=$?"; tail -n 8 gpurun_out/pytest_head.log
cat gpurun_out/parity_report.jsonl
for b in 1 4; do timeout 300 python tools/head_timing.py $b bf16 2>&1 | tail -4; done
timeout 300 python tools/head_timing.py 1 fp32 2>&1 | tail -4
