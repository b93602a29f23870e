#!/bin/bash
# Round 6, call 23: the backward in DD_PREC_F16X3 (split-f16 forward, fp32 gradient kernels): the GPU backward tests, then the training step of both heads
# in fp32 / f16x3 / bf16 on one box
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_backward.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06_call23_pytest_backward.txt
cp gpurun_out/parity_report.jsonl gpurun_out/r06_call23_parity_report.jsonl 2>/dev/null
for V in res swin; do for P in fp32 f16x3 bf16; do
  echo "== $V $P"; timeout 900 python bench.py --mode train-dp --variant $V --precision $P --batch 4 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print(o['value'], o['unit'], o['ms_per_step'], 'ms/step')
    elif 'Error' in l or 'error' in l: print(l.rstrip()[:300])
"
done; done > gpurun_out/r06_call23_train_f16x3.txt 2>&1
cat gpurun_out/r06_call23_pytest_backward.txt gpurun_out/r06_call23_train_f16x3.txt
