#!/bin/bash
# concurrent lanes (dd_set_option "streams"): GPU tests, then the bench line with its two_streams extra, B = 4 / 8 / 16, NYU, Swin
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fpn.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 5
X="--no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1"
for B in 4 8 16; do timeout 300 python bench.py --steps 10 --warmup 2 --batch $B $X 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B', d['value'], d['two_streams'])"; done
timeout 300 python bench.py --steps 10 --warmup 2 --size nyu $X 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nyu', d['value'], d['two_streams'])"
timeout 300 python bench.py --steps 5 --warmup 2 --variant swin $X 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('swin', d['value'], d['two_streams'])"
timeout 300 python bench.py --steps 10 --warmup 2 --streams 2 $X 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('--streams 2', d['value'], d['ms_per_step'], d['roofline']['loop_ms_graph'], d['roofline']['frac'])"
