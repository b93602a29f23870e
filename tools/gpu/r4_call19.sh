#!/bin/bash
# Round 4, call 19: A/B of the current library against build_variants/libddepth_base.so (the previous commit's build): loop / per-layer times, f16r
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
export DD_PRECS=${DD_PRECS:-f16r}
for lib in build_variants/libddepth_base.so diffusiondepth_amd/libddepth_hip.so build_variants/libddepth_base.so diffusiondepth_amd/libddepth_hip.so; do
  echo "== $lib"
  DDEPTH_LIBRARY=$PWD/$lib timeout 400 python tools/variant_bench.py 4 1 2>&1 | grep "^\["
done
