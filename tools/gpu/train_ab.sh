#!/bin/bash
# training step after a change: GPU backward tests, then the train-dp bench lines (Swin + Res heads), then the loop fwd+bwd timing tool
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_zz_gpu_heads.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 8
for V in swin res; do
  timeout 400 python bench.py --mode train-dp --variant $V --batch 4 --steps 3 --warmup 1 > gpurun_out/train_dp_${V}_b4.log 2>&1; tail -n 1 gpurun_out/train_dp_${V}_b4.log | cut -c1-330
done
timeout 300 python tools/train_step_timing.py 1 20 bf16 2>&1 | grep -v "^$" | tail -n 5
