#!/bin/bash
# Round 6, call 6: the skeleton with a lean weight-DMA issue sequence (MODE bit 4096); the reference-facade GPU test (both families).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 build_variants/conv_skeleton 0.4 2>&1 | tail -8 > gpurun_out/r06_call6_skeleton_lean.txt
timeout 900 python -m pytest tests/test_reference_facade.py -q -m gpu > gpurun_out/r06_call6_facade.txt 2>&1
cat gpurun_out/r06_call6_skeleton_lean.txt; tail -8 gpurun_out/r06_call6_facade.txt
