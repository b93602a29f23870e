#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--steps 20 --warmup 3 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-streams-extra"
for o in big_tiles=0 big_tiles=-1 big_tiles=0 big_tiles=-1; do
  for B in 4 8; do echo "== $o B=$B"; timeout 300 python bench.py $X --batch $B --set $o 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_us'], d['roofline']['loop_ms_graph'], d.get('latency_b1'))"; done
done
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 4
