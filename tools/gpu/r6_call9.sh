#!/bin/bash
# Round 6, call 9: the 16-bit training step's intermittent non-finite gradients OUTSIDE torch (tools/micro/train_graph_repro.cpp): graph replay vs eager
# trajectory-keeping forward, on the legacy NULL stream / a created blocking stream / a created non-blocking stream; 6 processes per arm, alternating.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
R=build_variants/train_graph_repro
{
for rep in 1 2 3 4 5 6; do
  for smode in 0 1 2; do
    for tg in 1 0; do
      timeout 120 $R $tg $smode 40 2 4 176 608 20 1 2>&1 | grep -v "amdgpu.ids" | tail -n 1
    done
  done
done
echo "== two lanes, bf16"
for rep in 1 2 3 4; do for tg in 1 0; do timeout 120 $R $tg 0 40 2 4 176 608 20 2 2>&1 | grep -v "amdgpu.ids" | tail -n 1; done; done
echo "== f16"
for rep in 1 2 3 4; do for tg in 1 0; do timeout 120 $R $tg 0 40 3 4 176 608 20 1 2>&1 | grep -v "amdgpu.ids" | tail -n 1; done; done
} > gpurun_out/r06_call9_train_graph_repro.txt 2>&1
grep -c FAILED gpurun_out/r06_call9_train_graph_repro.txt; grep -c clean gpurun_out/r06_call9_train_graph_repro.txt; grep FAILED gpurun_out/r06_call9_train_graph_repro.txt | head -20; head -4 gpurun_out/r06_call9_train_graph_repro.txt
