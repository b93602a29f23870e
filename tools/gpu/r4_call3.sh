#!/bin/bash
# Round 4, call 3: hoisted conv3 on 8x32 tiles with ONE patch buffer = three workgroups per CU (-DDD_C3_NPB1=1) against the default library,
# with the 16x32 tiles (default at B=4) and with 8x32 tiles forced, bf16 and f16r, one stream and two lanes; grid barrier probe 2 (fixed polling).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
export DD_PRECS=bf16,f16r
echo "== default library"
DDEPTH_LIBRARY=$PWD/diffusiondepth_amd/libddepth_hip.so timeout 400 python tools/variant_bench.py 4 1 2>&1 | grep "^\["
echo "== default library, 8x32 tiles forced"
DD_OPTS=big_tiles=0 DDEPTH_LIBRARY=$PWD/diffusiondepth_amd/libddepth_hip.so timeout 400 python tools/variant_bench.py 4 2>&1 | grep "^\["
echo "== npb1, 8x32 tiles forced (three workgroups per CU)"
DD_OPTS=big_tiles=0 DDEPTH_LIBRARY=$PWD/build_variants/libddepth_npb1.so timeout 400 python tools/variant_bench.py 4 1 2>&1 | grep "^\["
echo "== npb1, automatic tile rule"
DDEPTH_LIBRARY=$PWD/build_variants/libddepth_npb1.so timeout 400 python tools/variant_bench.py 4 2>&1 | grep "^\["
echo "== grid barrier probe 2"
timeout 100 build_variants/grid_sync_probe2 > gpurun_out/grid_sync_probe2b.txt 2>&1; echo "rc=$?"; cat gpurun_out/grid_sync_probe2b.txt
