#!/bin/bash
# Round 6, call 12: the graph-replay fault needs MIOpen (call 11: head 8 / 12, the same head with cudnn.enabled = False 0 / 12, library alone 0 / 12).  Which MIOpen
# operation?  Convolutions only / batch norms only / batch norms in eval mode; and the eager arm (train_graphs = 0) with MIOpen as the control.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in $(seq 1 10); do
  for arm in head head_bn_native head_conv_native head_eval_bn; do
    timeout 300 python tools/nan_arms.py $arm 1 40 2>&1 | grep "^\[" | tail -n 1
  done
  timeout 300 python tools/nan_arms.py head 0 40 2>&1 | grep "^\[" | tail -n 1
done > gpurun_out/r06_call12_nan_arms.txt 2>&1
for arm in "head tg=1" "head_bn_native tg=1" "head_conv_native tg=1" "head_eval_bn tg=1" "head tg=0"; do echo "$arm: $(grep "^\[$arm\]" gpurun_out/r06_call12_nan_arms.txt | grep -vc 'bad iterations: 0') failing of $(grep -c "^\[$arm\]" gpurun_out/r06_call12_nan_arms.txt)"; done | tee -a gpurun_out/r06_call12_nan_arms.txt
