#!/bin/bash
# (record of a finished experiment: "new" = the library built with tools/micro/r05_store_addr.patch applied, "base" = a copy of the unpatched build under build_variants/libddepth_r5base.so)
# Round 5, call 28: epilogue store addresses split into a wave-uniform block / plane term (scalar) and a per-pixel-row term (the compiler spent 14 VALU with
# 64-bit multiplies on every 16-byte store: conv2 -127 VALU per split) -- bit-identical results; the in-tree library against the round's baseline library.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DD_PRECS=f16r,bf16
run() { echo "== $1"; DDEPTH_LIBRARY=$2 timeout 300 python tools/variant_bench.py 4 2>&1 | grep -v "amdgpu.ids" | tail -n 3; }
{
run base build_variants/libddepth_r5base.so
run new diffusiondepth_amd/libddepth_hip.so
run base build_variants/libddepth_r5base.so
run new diffusiondepth_amd/libddepth_hip.so
run base build_variants/libddepth_r5base.so
run new diffusiondepth_amd/libddepth_hip.so
} > gpurun_out/call28_store_addr.txt 2>&1
cat gpurun_out/call28_store_addr.txt
