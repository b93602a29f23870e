#!/bin/bash
# Round 6, call 10: does the round-5 torch harness (training forward as a replayed hipGraph: train_graphs = 1) still fail on today's box and today's library?
# Beside it the torch-free driver in its torch-like form (ddim_loss call between forward and backward, backward from a second host thread, syncs).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DDEPTH_GRAD_GUARD=0
sed -n '/^cat > \/tmp\/tl8.py/,/^PY$/p' tools/gpu/r5_call19.sh | sed '1d;$d' > /tmp/tl8.py
run() { timeout 900 python /tmp/tl8.py "$@" 2>&1 | grep "^\[" | grep -v "non-finite LOSS\|non-finite grads" | tail -n 1; }
R=build_variants/train_graph_repro
{
for i in 1 2 3 4 5 6; do
  FIXSEED=320 run res bf16 1 40 train_graphs=1
  timeout 200 $R 1 0 40 2 4 176 608 20 1 1 2>&1 | grep -v "amdgpu.ids" | tail -n 1 | cut -c1-260
  FIXSEED=320 run res bf16 1 40 none
  timeout 200 $R 1 0 40 2 4 176 608 20 2 1 2>&1 | grep -v "amdgpu.ids" | tail -n 1 | cut -c1-260
done
} > gpurun_out/r06_call10_nan_harness.txt 2>&1
cat gpurun_out/r06_call10_nan_harness.txt
