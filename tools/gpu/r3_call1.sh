#!/bin/bash
# Round 3, first GPU contact of the split-f16 mode: probe, the whole GPU suite, the new bench line, the abs-clean figure, and two A/Bs.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== f16 denorm probe"; timeout 60 build_variants/f16_denorm_probe 2>&1 | tee gpurun_out/f16_denorm_probe.txt
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 25 gpurun_out/pytest_gpu.log
echo "== bench C3 default"; timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_c3_bf16_b4.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_c3_bf16_b4.log | cut -c1-3000
echo "== bench C3 f16x3"; timeout 300 python bench.py --steps 5 --warmup 2 --precision f16x3 --no-train-extra --no-nlspn-extra --no-head-extra > gpurun_out/bench_c3_f16x3_b4.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_c3_f16x3_b4.log | cut -c1-2500
for B in 6 8; do echo "== bench C3 B=$B"; timeout 300 python bench.py --steps 10 --warmup 2 --batch $B --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 > gpurun_out/bench_c3_bf16_b$B.log 2>&1; tail -n 1 gpurun_out/bench_c3_bf16_b$B.log | cut -c1-700; done
echo "== variants"; for v in default cadd16; do
  if [ $v = default ]; then unset DDEPTH_LIBRARY; else export DDEPTH_LIBRARY=build_variants/libddepth_$v.so; fi
  timeout 300 python tools/variant_bench.py 4 1 2>&1 | tail -n 4; done; unset DDEPTH_LIBRARY
