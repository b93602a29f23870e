#!/bin/bash
# Round 6, call 7: the new default (8x32 one-buffer conv3 under lanes) through bench.py, at B = 4 / 6 / 8 / 16 (two lanes) and B = 4 with big_tiles = 1 (the old rule).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-latency-b1 --no-train-extra --no-head-extra --no-nlspn-extra --no-abs-extra --no-streams-extra"
for cfg in "4:" "4:--set big_tiles=1" "6:" "8:" "16:" "4:" "4:--set big_tiles=1"; do
  b=${cfg%%:*}; extra=${cfg#*:}
  echo "== B=$b $extra"
  timeout 600 python bench.py --batch $b $Q $extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print(o['value'], 'maps/s', o['ms_per_step'], 'ms/step; step frac', r['step_frac_of_peak'], '; one-stream per-layer us', r['per_layer_avg_us'], 'regions', o['spread']['timed_regions_maps_per_s'])
"
done > gpurun_out/r06_call7_batches.txt 2>&1
cat gpurun_out/r06_call7_batches.txt
