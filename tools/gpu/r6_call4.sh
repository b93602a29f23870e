#!/bin/bash
# Round 6, call 4: (a) timing ablation of today's kernels at KITTI B=4, f16r and bf16 (what does the weight DMA / the staging / the stores cost now?);
# (b) the reference-facade drop-in test on the GPU.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for prec in f16r bf16; do
  echo "== ablation $prec B=4"
  DDEPTH_LIBRARY=build_variants/libddepth_abl.so timeout 600 python tools/ablate.py $prec 4 2>&1 | grep -v "amdgpu.ids"
done
} > gpurun_out/r06_call4_ablate.txt 2>&1
timeout 900 python -m pytest tests/test_reference_facade.py -x -q -m gpu > gpurun_out/r06_call4_facade.txt 2>&1
cat gpurun_out/r06_call4_ablate.txt; tail -15 gpurun_out/r06_call4_facade.txt
