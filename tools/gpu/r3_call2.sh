#!/bin/bash
# Round 3, call 2: GPU suite with the new training-trajectory / budget tests; kernel variants (one box, A/B); lane counts.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 40 gpurun_out/pytest_gpu.log | cut -c1-400
echo "== variants"; for v in default cadd16 c4k16 occ8 spread default; do
  if [ $v = default ]; then unset DDEPTH_LIBRARY; else export DDEPTH_LIBRARY=build_variants/libddepth_$v.so; fi
  timeout 300 python tools/variant_bench.py 4 1 2>&1 | grep -v amdgpu.ids | tail -n 3; done; unset DDEPTH_LIBRARY
for S in 1 2 4; do for B in 4 8; do echo "== bench streams=$S B=$B"; timeout 300 python bench.py --steps 10 --warmup 2 --batch $B --streams $S --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra 2>/dev/null | tail -n 1 | cut -c1-260; done; done
