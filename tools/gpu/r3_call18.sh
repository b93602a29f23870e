#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for o in thin_stream=0 thin_stream=1 thin_stream=0 thin_stream=1; do echo "== $o"; DD_OPTS=$o timeout 300 python tools/variant_bench.py 4 8 1 2>&1 | grep -v amdgpu.ids | tail -n 4; done
bash tools/gpu/r3_stamp.sh
