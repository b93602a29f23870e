#!/bin/bash
# Round 4, call 24: the NCHW-reading conv3(cond) with its raw patch two chunks ahead (RAW_DEPTH 2) against the previous build: once-per-image kernel time and the step
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "explicit_condition_tensor" 2>&1 | tail -2
for lib in build_variants/libddepth_base.so diffusiondepth_amd/libddepth_hip.so build_variants/libddepth_base.so diffusiondepth_amd/libddepth_hip.so; do
  export DDEPTH_LIBRARY=$R/$lib
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_l8" -o bench --output-format csv -- python "$R/bench.py" --steps 5 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra --no-abs-extra > "$R/gpurun_out/rocprof_l8.log" 2>&1)
  KS=$(find gpurun_out/prof_l8 -name "*kernel_stats.csv" | head -1)
  echo "== $lib: $(grep 'Cfg2<4, 47>' $KS | cut -d, -f2-4)   step: $(tail -n 1 gpurun_out/rocprof_l8.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  rm -rf gpurun_out/prof_l8
done
