#!/bin/bash
# Round 5, call 22: lane count at small image sizes (NYU: 75 tiles per image do not fill 512 slots)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--no-train-extra --no-nlspn-extra --no-head-extra --no-cpu-baseline --no-latency-b1 --no-streams-extra"
line() { name=$1; shift; timeout 600 python bench.py "$@" > gpurun_out/bench_$name.log 2>&1; echo "== $name rc=$? $(tail -n 1 gpurun_out/bench_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['spread']['timed_regions_maps_per_s'])")"; }
for i in 1 2; do
  for s in 1 2 3 4; do line nyu_b4_s${s}_$i --size nyu --batch 4 --streams $s --steps 20 --warmup 3 $X; done
  for s in 2 4; do line nyu_b8_s${s}_$i --size nyu --batch 8 --streams $s --steps 20 --warmup 3 $X; done
  for s in 2 4; do line nyu_b16_s${s}_$i --size nyu --batch 16 --streams $s --steps 10 --warmup 3 $X; done
done
