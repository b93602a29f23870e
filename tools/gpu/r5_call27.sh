#!/bin/bash
# Round 5, call 27: the two-lane line with the hoisted conv3 on 16x32 tiles (automatic under lanes) against the 8x32 one-buffer form (big_tiles=0), through bench.py
# (call 26's variant_bench.py -- denoise calls back to back, no codec -- read 538 vs 568 maps/s; call 3's bench.py read them equal)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--no-train-extra --no-nlspn-extra --no-head-extra --no-cpu-baseline --no-latency-b1 --no-streams-extra --no-abs-extra"
line() { name=$1; shift; timeout 300 python bench.py "$@" > gpurun_out/bench_$name.log 2>&1; echo "== $name rc=$?"; tail -n 1 gpurun_out/bench_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['spread']['timed_regions_maps_per_s'])"; }
{
for i in 1 2 3; do
  line auto_$i --steps 20 --warmup 5 $X
  line small_$i --steps 20 --warmup 5 --set big_tiles=0 $X
done
} 2>&1 | tee gpurun_out/call27_big_tiles.txt
