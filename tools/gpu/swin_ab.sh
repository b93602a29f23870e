#!/bin/bash
# A/B of library variants on the Swin denoiser: parity tests of the Swin paths, then the bench line (one stream) twice per variant, alternating
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra"
for n in "$@"; do
  DDEPTH_LIBRARY=$PWD/build_variants/libddepth_$n.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -m gpu -q -x -k "swin" -p no:cacheprovider 2>&1 | tail -n 2
done
for rep in 1 2; do for n in "$@"; do
  DDEPTH_LIBRARY=$PWD/build_variants/libddepth_$n.so timeout 300 python bench.py --variant swin --steps 5 --warmup 2 $X 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], 'maps/s  loop', d['roofline']['loop_ms_graph'], 'ms', d['roofline']['per_layer_avg_us'])"
done; done
