#!/bin/bash
# A/B: prologue load order (table inputs -> patch -> accumulator start values, no wait in between) against the previous build
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2; do
for lib in build_variants/libddepth_head.so diffusiondepth_amd/libddepth_hip.so; do echo "== $lib"; DDEPTH_LIBRARY=$lib timeout 300 python tools/variant_bench.py 4 1 2>&1 | grep -v amdgpu.ids | tail -n 7; done
done
echo "== phase profile"; timeout 300 python tools/phase_prof.py run 9,2 4 bf16 2>&1 | grep -v amdgpu.ids | tail -n 26
