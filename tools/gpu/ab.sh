#!/bin/bash
# A/B/A/B of library variants / option sets on ONE box (box-to-box spread is +-5 %: never compare across gpurun calls).
#   tools/gpu/ab.sh <out-name> <arm> [<arm> ...]       arm = "lib=<variant .so>" and / or "opts=<k=v,k=v>" joined by ';', or "default"
#   e.g.  gpurun -- 'bash tools/gpu/ab.sh r06_bigtiles default "opts=big_tiles=1" default "opts=big_tiles=1"'
#         gpurun -- 'bash tools/gpu/ab.sh r06_wreg default "lib=build_variants/libddepth_wreg.so" default "lib=build_variants/libddepth_wreg.so"'
# Environment: DD_PRECS (default f16r), AB_BATCHES (default "4").  Every arm runs tools/variant_bench.py: parity on ragged shapes against the fp64 oracle,
# then loop / per-layer times at KITTI size, one stream and two lanes.  Output: gpurun_out/<out-name>.txt
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DD_PRECS=${DD_PRECS:-f16r}
out=gpurun_out/$1.txt; shift
for arm in "$@"; do
  lib=diffusiondepth_amd/libddepth_hip.so; opts=""
  IFS=';' read -ra parts <<< "$arm"
  for p in "${parts[@]}"; do case "$p" in lib=*) lib=${p#lib=};; opts=*) opts=${p#opts=};; esac; done
  echo "== $arm"
  DDEPTH_LIBRARY=$lib DD_OPTS=$opts timeout 600 python tools/variant_bench.py ${AB_BATCHES:-4} 2>&1 | grep -v "amdgpu.ids" | tail -n 6
done > "$out" 2>&1
cat "$out"
