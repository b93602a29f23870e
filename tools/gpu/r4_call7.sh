#!/bin/bash
# Round 4, call 7: GroupNorm statistic slots private to an XCD (default build) against slots shared between XCDs (-DDD_STAT_XCD=0), one stream and two lanes
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
export DD_PRECS=bf16,f16r
echo "== slots shared between XCDs (round 3's rule)"
DDEPTH_LIBRARY=$PWD/build_variants/libddepth_statshared.so timeout 400 python tools/variant_bench.py 4 1 2>&1 | grep "^\["
echo "== slots private to an XCD"
DDEPTH_LIBRARY=$PWD/diffusiondepth_amd/libddepth_hip.so timeout 400 python tools/variant_bench.py 4 1 2>&1 | grep "^\["
