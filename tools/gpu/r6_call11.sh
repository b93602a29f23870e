#!/bin/bash
# Round 6, call 11: which part of the torch-side training step does the graph-replay fault need?  Four arms (tools/nan_arms.py), train_graphs = 1, interleaved,
# 12 processes each (the fault hit 1 of 6 head processes on call 10's box).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in $(seq 1 12); do
  for arm in head lib autograd head_nomiopen; do
    timeout 300 python tools/nan_arms.py $arm 1 40 2>&1 | grep "^\[" | tail -n 1
  done
done > gpurun_out/r06_call11_nan_arms.txt 2>&1
for arm in head lib autograd head_nomiopen; do echo "$arm: $(grep "^\[$arm " gpurun_out/r06_call11_nan_arms.txt | grep -vc 'bad iterations: 0') failing of $(grep -c "^\[$arm " gpurun_out/r06_call11_nan_arms.txt)"; done | tee -a gpurun_out/r06_call11_nan_arms.txt
grep -v "bad iterations: 0" gpurun_out/r06_call11_nan_arms.txt
