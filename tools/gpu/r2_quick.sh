#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/variant_bench.py 4 1 16 2>&1 | grep "^\["
timeout 300 python tools/bwd_timing.py 4 bf16 2>&1 | tail -1
timeout 400 python bench.py --mode train-dp --variant swin --batch 4 --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-330
