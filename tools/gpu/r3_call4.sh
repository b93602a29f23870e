#!/bin/bash
# Round 3, call 4: persistent hoisted conv3 (A/B through the option persist_slots), parity tests.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for o in persist_slots=0 persist_slots=512 persist_slots=768 persist_slots=0 persist_slots=512; do echo "== $o"; DD_OPTS=$o timeout 300 python tools/variant_bench.py 4 8 2 2>&1 | grep -v amdgpu.ids | tail -n 4; done
echo "== pytest parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 5
for S in 1 2; do echo "== bench streams=$S"; timeout 300 python bench.py --steps 20 --warmup 3 --streams $S --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra 2>/dev/null | tail -n 1 | cut -c1-1600; done
