#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 8 gpurun_out/pytest_gpu.log
echo "== phase prof B=4"; timeout 300 python tools/phase_prof.py run 2,9,3,1,4 4 bf16 2>&1 | grep -v amdgpu.ids
echo "== phase prof B=1"; timeout 300 python tools/phase_prof.py run 2,9,1,4 1 bf16 2>&1 | grep -v amdgpu.ids
