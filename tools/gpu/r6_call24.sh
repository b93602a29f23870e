#!/bin/bash
# Round 6, call 24: DD_PREC_F16X3's backward with f16 gradients behind the split forward: the GPU backward tests, then the training step of both heads in
# f16x3 / bf16 (alternating, one box)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests/test_gpu_backward.py -q -m gpu 2>&1 | tail -25 > gpurun_out/r06_call24_pytest_backward.txt
cp gpurun_out/parity_report.jsonl gpurun_out/r06_call24_parity_report.jsonl 2>/dev/null
for rep in 1 2; do for V in res swin; do for P in f16x3 bf16; do
  echo "== $V $P"; timeout 900 python bench.py --mode train-dp --variant $V --precision $P --batch 4 --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print(o['value'], o['unit'], o['ms_per_step'], 'ms/step')
    elif 'Error' in l or 'error' in l: print(l.rstrip()[:300])
"
done; done; done > gpurun_out/r06_call24_train_f16x3.txt 2>&1
cat gpurun_out/r06_call24_pytest_backward.txt gpurun_out/r06_call24_train_f16x3.txt
