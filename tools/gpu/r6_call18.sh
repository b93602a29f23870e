#!/bin/bash
# Round 6, call 18: what of the weight DMA costs?  ablation bits: 4 = no DMA, 4096 = DMA issued but never waited for, 8192 = half the pieces, 12288 = both
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
ABL_MASKS=0,4,4096,8192,12288,0,4,4096,8192 DDEPTH_LIBRARY=build_variants/libddepth_abl3.so timeout 600 python tools/ablate.py f16r 4 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r06_call18_ablate_dma.txt
cat gpurun_out/r06_call18_ablate_dma.txt
