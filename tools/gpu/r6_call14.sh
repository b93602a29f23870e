#!/bin/bash
# Round 6, call 14: is the training-graph fault in the HIP runtime's graph fast path?  The head arm (train_graphs = 1) under runtime switches:
# DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (graph nodes enqueued as ordinary commands instead of pre-captured AQL packets), HIP_FORCE_DEV_KERNARG=0 / 1, AMD_DIRECT_DISPATCH=0.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in $(seq 1 8); do
  timeout 300 python tools/nan_arms.py head 1 40 2>&1 | grep "^\[" | tail -n 1
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python tools/nan_arms.py head 1 40 2>&1 | grep "^\[" | tail -n 1
  HIP_FORCE_DEV_KERNARG=0 timeout 300 python tools/nan_arms.py head 1 40 2>&1 | grep "^\[" | tail -n 1
  HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/nan_arms.py head 1 40 2>&1 | grep "^\[" | tail -n 1
  AMD_DIRECT_DISPATCH=0 timeout 300 python tools/nan_arms.py head 1 40 2>&1 | grep "^\[" | tail -n 1
done > gpurun_out/r06_call14_nan_runtime_switches.txt 2>&1
for arm in "head tg=1]" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0]" "HIP_FORCE_DEV_KERNARG=0]" "HIP_FORCE_DEV_KERNARG=1]" "AMD_DIRECT_DISPATCH=0]"; do echo "$arm: $(grep -F "$arm" gpurun_out/r06_call14_nan_runtime_switches.txt | grep -vc 'bad iterations: 0') failing of $(grep -cF "$arm" gpurun_out/r06_call14_nan_runtime_switches.txt)"; done | tee -a gpurun_out/r06_call14_nan_runtime_switches.txt
head -6 gpurun_out/r06_call14_nan_runtime_switches.txt | cut -c1-220
