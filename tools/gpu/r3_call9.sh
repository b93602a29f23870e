#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== codec timing"; timeout 300 python tools/codec_timing.py 2>&1 | grep -v amdgpu.ids
echo "== pytest codec + parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 3
echo "== bench"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra 2>/dev/null | tail -n 1 | cut -c1-330
