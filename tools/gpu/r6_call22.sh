#!/bin/bash
# Round 6, call 22: what the gradient guard costs the training step: train-dp lines with the (non-blocking) guard and with DDEPTH_GRAD_GUARD=0, alternating
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for g in 1 0; do for V in res swin; do
  echo "== guard=$g $V"; DDEPTH_GRAD_GUARD=$g timeout 600 python bench.py --mode train-dp --variant $V --batch 4 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print(o['value'], o['unit'], o['ms_per_step'], 'ms/step')
"
done; done; done > gpurun_out/r06_call22_train_guard.txt 2>&1
cat gpurun_out/r06_call22_train_guard.txt
