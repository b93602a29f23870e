#!/bin/bash
# Round 6, call 3: conv1 as persistent workgroups on the dynamic tile ticket (option walk_conv1), with and without the walks of conv2 / conv3; big_tiles = 0 under lanes.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DD_PRECS=f16r
run() { echo "== $1"; DD_OPTS=$1 timeout 300 python tools/variant_bench.py 4 2>&1 | grep -v "amdgpu.ids" | tail -n 2; }
{
run big_tiles=0
run walk_conv1=512,big_tiles=0
run walk_conv1=768,big_tiles=0
run walk_conv1=1024,big_tiles=0
run walk_conv1=256,big_tiles=0
run walk_conv1=512,walk_conv2=512,walk_conv3=768,big_tiles=0
run big_tiles=0
run ""
} > gpurun_out/r06_call3_walk1.txt 2>&1
cat gpurun_out/r06_call3_walk1.txt
