#!/bin/bash
# Round 4, call 6: where does the time of the thin kernels go?  Ablation (-DDD_ABLATE=1 build) of conv1 (general kernel) and conv4 (streaming kernel), f16r and bf16, B=4
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
export DDEPTH_LIBRARY=$PWD/build_variants/libddepth_ablate.so
for prec in f16r bf16; do
  echo "== $prec B=4"
  ABL_MASKS=0,1,2,3,8,16,32,48,64,128,24,27,256,512,1024,0 timeout 300 python tools/ablate.py $prec 4 2>&1 | grep "^ablate"
done
