#!/bin/bash
# Swin hoist, 5x5 form: GPU parity (A/B test + the Swin tests), then the Swin bench lines with the two hoisted forms
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "swin" 2>&1 | tail -n 8
grep swin_hoist_ab gpurun_out/parity_report.jsonl | cut -c1-330
for w5 in 0 1; do for prec in f16 bf16; do
echo "== swin $prec swin_w5=$w5"; timeout 400 python bench.py --steps 5 --warmup 2 --variant swin --precision $prec --set swin_w5=$w5 --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 2>&1 | tail -n 1 | cut -c1-1500
done; done
