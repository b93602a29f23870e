#!/bin/bash
# Round 6, call 5: is it the fp64 ATOMICS of the statistics epilogue (ablation bit 32 skips them: -13 % on conv2)?  bit 2048 = a plain store in their place.
# Then the reference-facade test again (synthetic head weights).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for prec in f16r; do
  echo "== ablation $prec B=4 (bit 32: no statistics atomics; bit 2048: plain stores instead)"
  ABL_MASKS=0,32,2048,0,32,2048,16,2064,0 DDEPTH_LIBRARY=build_variants/libddepth_abl2.so timeout 600 python tools/ablate.py $prec 4 2>&1 | grep -v "amdgpu.ids"
  echo "== B=1"
  ABL_MASKS=0,32,2048,0 DDEPTH_LIBRARY=build_variants/libddepth_abl2.so timeout 600 python tools/ablate.py $prec 1 2>&1 | grep -v "amdgpu.ids"
done
} > gpurun_out/r06_call5_ablate_atomics.txt 2>&1
timeout 900 python -m pytest tests/test_reference_facade.py -x -q -m gpu > gpurun_out/r06_call5_facade.txt 2>&1
cat gpurun_out/r06_call5_ablate_atomics.txt; tail -15 gpurun_out/r06_call5_facade.txt
