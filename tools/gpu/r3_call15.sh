#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--steps 20 --warmup 3 --size nyu --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-streams-extra --no-latency-b1"
for B in 4 16; do for S in 1 2 4; do echo "== nyu B=$B streams=$S"; timeout 300 python bench.py $X --batch $B --streams $S 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['step_frac_of_peak'])"; done; done
