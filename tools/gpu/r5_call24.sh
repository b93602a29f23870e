#!/bin/bash
# Round 5, call 24: streaming conv4 walking CONSECUTIVE tiles per workgroup (left halo from the previous tile, workgroups spread over the image) vs the strided walk
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { name=$1; lib=$2; echo "== $name"; DDEPTH_LIBRARY=$lib DD_PRECS=f16r,bf16 timeout 300 python tools/variant_bench.py 4 1 > gpurun_out/variant_$name.log 2>&1; echo "rc=$?"; grep -v "^$" gpurun_out/variant_$name.log | grep "parity\|B=" | cut -c1-330; }
D=diffusiondepth_amd/libddepth_hip.so
run default_a $D
run c4contig_a build_variants/libddepth_c4contig.so
run default_b $D
run c4contig_b build_variants/libddepth_c4contig.so
