#!/bin/bash
# Round 3, call 8: conv1 as a persistent streaming kernel too (thin_stream 0 / 1 / 2), parity + backward tests.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for o in thin_stream=0 thin_stream=1 thin_stream=2 thin_stream=1 thin_stream=2; do echo "== $o"; DD_OPTS=$o timeout 300 python tools/variant_bench.py 4 1 8 2>&1 | grep -v amdgpu.ids | tail -n 4; done
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 5
echo "== bench"; timeout 300 python bench.py --steps 20 --warmup 3 --no-train-extra --no-nlspn-extra --no-head-extra 2>/dev/null | tail -n 1 | cut -c1-2000
