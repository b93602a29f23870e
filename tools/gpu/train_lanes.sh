#!/bin/bash
# training with concurrent lanes: GPU backward tests, then the train-dp bench lines with DDEPTH_STREAMS=1 / 2 (Swin and Res heads), alternating
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 6
for rep in 1 2; do for S in 1 2; do for V in swin res; do
  DDEPTH_STREAMS=$S timeout 400 python bench.py --mode train-dp --variant $V --batch 4 --steps 3 --warmup 1 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams=$S $V', d['value'], 'samples/s', d['ms_per_step'], 'ms')"
done; done; done
