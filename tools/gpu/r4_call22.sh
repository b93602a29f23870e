#!/bin/bash
# Round 4, call 22: the streaming conv4 on 16x32 tiles (two rows per wave; option thin_tall) against its 8-row form and against the previous build
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
export DD_PRECS=${DD_PRECS:-f16r,bf16}
run() { echo "== $1 opts=$2"; DDEPTH_LIBRARY=$PWD/$1 DD_OPTS=$2 timeout 400 python tools/variant_bench.py 4 2 8 2>&1 | grep "^\["; }
for rep in 1 2; do
  run build_variants/libddepth_base.so ""
  run diffusiondepth_amd/libddepth_hip.so thin_tall=2
  run diffusiondepth_amd/libddepth_hip.so thin_tall=1
done
