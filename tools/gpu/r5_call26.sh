#!/bin/bash
# (record of a finished experiment: the options exist only with tools/micro/r05_tile_walk.patch applied -- `git apply tools/micro/r05_tile_walk.patch && python -m diffusiondepth_amd.build`)
# Round 5, call 26: conv2 / the hoisted conv3 as tile-walking workgroups (options walk_conv2 / walk_conv3: tables once per workgroup, B x n workgroups) against one tile
# per workgroup, on one stream and under two lanes (there with the 8x32 one-buffer conv3: the 16x32 form spills with the tile loop); KITTI B=4, f16r.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DD_PRECS=f16r DD_CMP=1
run() { echo "== DD_OPTS=$1"; DD_OPTS=$1 timeout 300 python tools/variant_bench.py 4 2>&1 | grep -v "^$" | tail -n 4; }
{
run ""
run "walk_conv2=512"
run "walk_conv3=768"
run "walk_conv2=512,walk_conv3=768"
run "big_tiles=0"
run "big_tiles=0,walk_conv2=512,walk_conv3=768"
run "big_tiles=0,walk_conv2=256,walk_conv3=384"
run "big_tiles=0,walk_conv2=512,walk_conv3=512"
run ""
run "big_tiles=0,walk_conv2=512,walk_conv3=768"
} > gpurun_out/call26_walk.txt 2>&1
cat gpurun_out/call26_walk.txt
