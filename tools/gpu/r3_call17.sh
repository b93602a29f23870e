#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--steps 20 --warmup 3 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-streams-extra --no-latency-b1"
for cfg in "--streams 2" "--streams 4" "--streams 4 --set big_tiles=1" "--streams 3" "--streams 2"; do for B in 4 8; do echo "== B=$B $cfg"; timeout 300 python bench.py $X --batch $B $cfg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
