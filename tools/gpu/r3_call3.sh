#!/bin/bash
# Round 3, call 3: eight-wave latency variants of conv2 / hoisted conv3 at B = 1 (A/B through the option latency_tiles), NYU sizes.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for o in latency_tiles=0 latency_tiles=512 latency_tiles=1024 latency_tiles=0 latency_tiles=512; do echo "== $o"; DD_OPTS=$o timeout 300 python tools/variant_bench.py 1 2 2>&1 | grep -v amdgpu.ids | tail -n 3; done
echo "== pytest parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 5
echo "== bench B=1"; timeout 300 python bench.py --steps 20 --warmup 3 --batch 1 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra 2>/dev/null | tail -n 1 | cut -c1-1800
echo "== bench nyu"; timeout 300 python bench.py --steps 20 --warmup 3 --size nyu --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra 2>/dev/null | tail -n 1 | cut -c1-1200
