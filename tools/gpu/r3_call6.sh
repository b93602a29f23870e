#!/bin/bash
# Round 3, call 6: conv4 as the persistent streaming kernel (dd_thin.hip), A/B through the option thin_stream; slots sweep.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for o in thin_stream=0 thin_stream=1 thin_slots=768 thin_slots=1024 thin_stream=0 thin_stream=1; do echo "== $o"; DD_OPTS=$o timeout 300 python tools/variant_bench.py 4 1 8 2>&1 | grep -v amdgpu.ids | tail -n 4; done
echo "== pytest parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 5
