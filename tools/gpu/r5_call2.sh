#!/bin/bash
# Round 5, call 2: library variants (tools/build_variant.py -> build_variants/) A/B on one box: row-reuse MFMA block of the conv3-shaped layers (DD_DY_REUSE),
# conv1 on the interleaved workgroup -> tile map (DD_FLAT_L1), conv2 with three fragment register buffers (DD_FD_CONV2=3)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { name=$1; lib=$2; echo "== $name"; DDEPTH_LIBRARY=$lib DD_PRECS=f16r,bf16 timeout 300 python tools/variant_bench.py 4 1 > gpurun_out/variant_$name.log 2>&1; echo "rc=$?"; grep -v "^$" gpurun_out/variant_$name.log | tail -n 6 | cut -c1-330; }
D=diffusiondepth_amd/libddepth_hip.so
run default_a $D
run dy build_variants/libddepth_dy.so
run flat build_variants/libddepth_flat.so
run fd3 build_variants/libddepth_fd3.so
run dyflat build_variants/libddepth_dyflat.so
run default_b $D
run dy_b build_variants/libddepth_dy.so
