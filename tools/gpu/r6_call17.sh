#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/gpu/dbg_swin_once.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_call17_dbg.txt; cat gpurun_out/r06_call17_dbg.txt
export DD_PRECS=f16r,bf16
run() { echo "== $1"; DDEPTH_LIBRARY=$2 timeout 300 python tools/variant_bench.py 4 2>&1 | grep -v "amdgpu.ids" | tail -n 3; }
{
run base diffusiondepth_amd/libddepth_hip.so
run preissue build_variants/libddepth_preissue.so
run base diffusiondepth_amd/libddepth_hip.so
run preissue build_variants/libddepth_preissue.so
} > gpurun_out/r06_call17_preissue.txt 2>&1
cat gpurun_out/r06_call17_preissue.txt
