#!/bin/bash
# Round 6, call 1: (a) the conv2 skeleton with the two workgroups of a CU out of step by construction (tools/micro/conv_skeleton.hip, MODE bit 2048);
# (b) the shipped kernels with the first round's workgroups skewed (options skew2_us / skew3_us), one stream and two lanes, A/B/.../A.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DD_PRECS=f16r
timeout 300 build_variants/conv_skeleton 0.5 > gpurun_out/r06_call1_skeleton.txt 2>&1
run() { echo "== $1"; DD_OPTS=$1 timeout 300 python tools/variant_bench.py 4 2>&1 | grep -v "amdgpu.ids" | tail -n 2; }
{
run ""
run skew2_us=10,skew3_us=10
run skew2_us=20,skew3_us=13
run skew2_us=20,skew3_us=20
run skew2_us=30,skew3_us=30
run skew2_us=20
run skew3_us=13
run ""
} > gpurun_out/r06_call1_skew.txt 2>&1
cat gpurun_out/r06_call1_skeleton.txt gpurun_out/r06_call1_skew.txt
