#!/bin/bash
# Swin hoist: GPU parity (new A/B test + the Swin tests), then the Swin bench lines with and without it
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "swin" 2>&1 | tail -n 8
grep swin_hoist_ab gpurun_out/parity_report.jsonl | cut -c1-400
for hc in 0 -1; do for prec in bf16 f16; do
echo "== swin $prec hoist_cond=$hc"; timeout 400 python bench.py --steps 5 --warmup 2 --variant swin --precision $prec --set hoist_cond=$hc --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 2>&1 | tail -n 1 | cut -c1-900
done; done
