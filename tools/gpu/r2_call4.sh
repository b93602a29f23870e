#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest heads"; timeout 900 python -m pytest tests/test_zz_gpu_heads.py tests/test_gpu_fpn.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -12
echo "== bench swin f16 T=50 (C5 shape) B=1"; timeout 600 python bench.py --variant swin --precision f16 --T 50 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra 2>&1 | tail -1 > gpurun_out/bench_c5_swin_f16_t50_b1.json; cat gpurun_out/bench_c5_swin_f16_t50_b1.json | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['roofline']['per_layer_avg_us'], d['head_forward'], d['nlspn_refine']['module_forward_ms'] if d.get('nlspn_refine') else None)"
