#!/bin/bash
# Round 6, call 8: LDS fill rate per CU from an L2-resident image -- LDS-DMA vs register staging (tools/micro/lds_fill_rate.hip)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 build_variants/lds_fill_rate > gpurun_out/r06_call8_lds_fill_rate.txt 2>&1
cat gpurun_out/r06_call8_lds_fill_rate.txt
