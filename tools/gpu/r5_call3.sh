#!/bin/bash
# Round 5, call 3: under two lanes, does the one-buffer 8x32 conv3 (52 KB of LDS: leaves room for the other lane's conv1 / conv4 workgroups on a CU) beat the 16x32 form?
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
X="--no-train-extra --no-nlspn-extra --no-head-extra --no-cpu-baseline --no-latency-b1 --no-streams-extra"
line() { name=$1; shift; timeout 600 python bench.py "$@" > gpurun_out/bench_$name.log 2>&1; echo "== $name rc=$?"; tail -n 1 gpurun_out/bench_$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['spread']['timed_regions_maps_per_s'])"; }
for i in 1 2; do
  line auto_s2_$i --steps 10 --warmup 3 $X
  line small_s2_$i --steps 10 --warmup 3 --set big_tiles=0 $X
  line small_s3_$i --steps 10 --warmup 3 --set big_tiles=0 --streams 3 $X
  line small_s4_$i --steps 10 --warmup 3 --set big_tiles=0 --streams 4 $X
  line auto_b6_$i --steps 10 --warmup 3 --batch 6 $X
  line small_b6_s3_$i --steps 10 --warmup 3 --batch 6 --streams 3 --set big_tiles=0 $X
done
