#!/bin/bash
# Round 6, call 16: the whole GPU suite (no -x) after the tile-rule test fix
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_call16_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/r06_call16_pytest_gpu.txt
cp gpurun_out/parity_report.jsonl gpurun_out/r06_call16_parity_report.jsonl 2>/dev/null
