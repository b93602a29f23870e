#!/bin/bash
# sample sclk / power while the bench loop runs (is the chip power- or clock-limited under the MFMA kernels?)
cd "$(dirname "$0")"; mkdir -p gpurun_out; export TMPDIR=/tmp
export DDEPTH_LIBRARY=$PWD/build_variants/lib_std3.so
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -v "^=\|^$" | head -20
(timeout 120 python bench.py --steps 400 --warmup 3 --precision ${PREC:-bf16} --batch 4 --no-cpu-baseline > gpurun_out/bench_clock.log 2>&1) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -i "sclk\|power\|busy\|mclk\|fclk" | tr '\n' ' '; echo; sleep 1.5; done
wait $BP
grep "^{" gpurun_out/bench_clock.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['per_layer_avg_us'])"
