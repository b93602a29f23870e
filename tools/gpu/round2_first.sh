#!/bin/bash
# First GPU call of round 2 (about 2 GPU-minutes): does the double-buffered Winograd convB (dd_wino.hip v2, never run) work, and how fast is it?
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
# host-side items found at the end of round 1: device route of the parameter refresh, the ddim_loss stages
timeout 300 python -m pytest tests/test_zzz_gpu_device_weights.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/train_step_timing.py 1 20 bf16 2>&1 | tail -4
timeout 120 python tools/ddim_loss_timing.py 4 bf16 2>&1 | tail -12
timeout 120 python tools/gpu/wino_try.py 2>&1 | tail -40
DD_TEST_WINOGRAD=1 timeout 120 python -m pytest tests/test_zz_gpu_wino.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6
for wv in 0 2 3 4 5; do timeout 120 python bench.py --variant swin --precision f16 --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --winograd $wv $( [ "$wv" -ge 2 ] && echo --winograd-dma ) 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('winograd', d['config']['winograd'], d['value'], 'maps/s  loop_ms', r['loop_ms_graph'], 'per_layer_us', r['per_layer_avg_us'])"; done
# PMC passes of the Winograd kernel afterwards (LDS bank conflicts, VALU / MFMA / LDS busy; ~2 GPU-minutes):
#   BENCH_CFG="--precision f16 --batch 4 --size kitti --variant swin --winograd 2 --winograd-dma" bash tools/gpu/pmc.sh
