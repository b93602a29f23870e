#!/bin/bash
# Round 6, call 13: (a) conv2 with its weight stages through registers, two stages ahead (-DDD_W_REGS=1) against the default library, A/B/A/B;
# (b) the training-graph fault with a host synchronisation in front of / behind the graph launch, or the graph on the handle's own stream.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DD_PRECS=f16r,bf16
run() { echo "== $1"; DDEPTH_LIBRARY=$2 timeout 300 python tools/variant_bench.py 4 1 2>&1 | grep -v "amdgpu.ids" | tail -n 5; }
{
run base diffusiondepth_amd/libddepth_hip.so
run wreg build_variants/libddepth_wreg.so
run base diffusiondepth_amd/libddepth_hip.so
run wreg build_variants/libddepth_wreg.so
} > gpurun_out/r06_call13_wreg.txt 2>&1
cat gpurun_out/r06_call13_wreg.txt
for i in $(seq 1 8); do
  for f in "" 1 2 4; do
    GRAPH_FENCE=$f timeout 300 python tools/nan_arms.py head 1 40 2>&1 | grep "^\[" | tail -n 1
  done
done > gpurun_out/r06_call13_nan_fence.txt 2>&1
for arm in "head tg=1" "head fence=1 tg=1" "head fence=2 tg=1" "head fence=4 tg=1"; do echo "$arm: $(grep -F "[$arm]" gpurun_out/r06_call13_nan_fence.txt | grep -vc 'bad iterations: 0') failing of $(grep -cF "[$arm]" gpurun_out/r06_call13_nan_fence.txt)"; done | tee -a gpurun_out/r06_call13_nan_fence.txt
