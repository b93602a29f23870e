#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--steps 20 --warmup 3 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra"
for v in default c3_2 default c3_2; do
  if [ $v = default ]; then unset DDEPTH_LIBRARY; else export DDEPTH_LIBRARY=build_variants/libddepth_$v.so; fi
  for B in 4 8; do echo "== $v B=$B"; timeout 300 python bench.py $X --batch $B 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_us'], d['roofline']['loop_ms_graph'])"; done
done
