#!/bin/bash
# Round 2, GPU call 1: (a) the round-1 red test after the dd_denoise_once fix, (b) where the eval-time ddim_loss milliseconds go,
# (c) the Winograd options that had never run (decision: keep the faster parity-green one or delete dd_wino.hip), (d) a default bench line.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== device weights test"; timeout 300 python -m pytest tests/test_zzz_gpu_device_weights.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
echo "== ddim_loss timing"; timeout 200 python tools/ddim_loss_timing.py 4 bf16 2>&1 | tail -14
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_ddimloss" -o dl --output-format csv -- python "$OLDPWD/tools/ddim_loss_timing.py" 4 bf16 > "$OLDPWD/gpurun_out/rocprof_ddimloss.log" 2>&1); echo "rocprof rc=$?"
for f in $(find gpurun_out/prof_ddimloss -name "*kernel_stats.csv" | head -1); do head -n 30 "$f"; done
find gpurun_out/prof_ddimloss -name "*kernel_trace.csv" -size +20M -delete
echo "== wino"; timeout 400 python tools/gpu/wino_try.py 2>&1 | tail -50
DD_TEST_WINOGRAD=1 timeout 200 python -m pytest tests/test_zz_gpu_wino.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -6
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_bf16.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench_bf16.log
