#!/bin/bash
# Round 6, call 25: the multi-scale deformable attention operator / module / neck on the GPU (tests/test_zz_gpu_msda.py)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests/test_zz_gpu_msda.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r06_call25_pytest_msda.txt
cp gpurun_out/parity_report.jsonl gpurun_out/r06_call25_parity_report.jsonl 2>/dev/null
cat gpurun_out/r06_call25_pytest_msda.txt gpurun_out/r06_call25_parity_report.jsonl
