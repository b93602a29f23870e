#!/bin/bash
# Round 6, call 2: conv2 / the hoisted conv3 as persistent workgroups on a DYNAMIC per-XCD tile ticket (options walk_conv2 / walk_conv3), tables rebuilt
# only when the image changes; one stream and two lanes, default first and last.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DD_PRECS=f16r
run() { echo "== $1"; DD_OPTS=$1 timeout 300 python tools/variant_bench.py 4 2>&1 | grep -v "amdgpu.ids" | tail -n 2; }
{
run ""
run walk_conv2=512
run walk_conv3=768
run walk_conv2=512,walk_conv3=768
run walk_conv2=512,walk_conv3=768,big_tiles=0
run walk_conv2=512,walk_conv3=512
run big_tiles=0
run walk_conv2=256,walk_conv3=384,big_tiles=0
run ""
} > gpurun_out/r06_call2_walk.txt 2>&1
cat gpurun_out/r06_call2_walk.txt
