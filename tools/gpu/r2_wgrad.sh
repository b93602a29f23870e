#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest backward"; timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_zzz_gpu_device_weights.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8
echo "== bwd timing"; timeout 300 python tools/bwd_timing.py 4 bf16 2>&1 | tail -6
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_bwd" -o bw --output-format csv -- python "$OLDPWD/tools/bwd_timing.py" 4 bf16 > /dev/null 2>&1)
for f in $(find gpurun_out/prof_bwd -name "*kernel_stats.csv" | head -1); do head -n 14 "$f" | cut -c1-160; done
find gpurun_out/prof_bwd -name "*kernel_trace.csv" -delete
echo "== train-dp swin"; timeout 400 python bench.py --mode train-dp --variant swin --batch 4 --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-330
echo "== train-dp res"; timeout 400 python bench.py --mode train-dp --variant res --batch 4 --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-330
