#!/bin/bash
# Round 2, GPU call 2: the bf16 mode with f16 storage + hoisted condition term -- full parity suite with the new asserted depth gates,
# then bench lines (default, all-bf16 A/B, hoist off, f16) and the data-parallel training step at N=1 (BASELINE config 4 shape).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/pytest_gpu.log
echo "== smoke";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 6 gpurun_out/smoke.log
B="--steps 10 --warmup 2 --no-train-extra --no-nlspn-extra --no-head-extra"
echo "== bench default";   timeout 300 python bench.py $B > gpurun_out/bench_bf16.log 2>&1; echo "rc=$?"; tail -n 2 gpurun_out/bench_bf16.log
echo "== bench all-bf16";  timeout 300 python bench.py $B --bf16-storage --no-parity-gate > gpurun_out/bench_bf16_pure.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_bf16_pure.log
echo "== bench hoist 0";   timeout 300 python bench.py $B --hoist 0 > gpurun_out/bench_bf16_nohoist.log 2>&1; echo "rc=$?"; tail -n 2 gpurun_out/bench_bf16_nohoist.log
echo "== bench f16";       timeout 300 python bench.py $B --precision f16 > gpurun_out/bench_f16.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_f16.log
echo "== train-dp swin B=4"; timeout 400 python bench.py --mode train-dp --variant swin --batch 4 --steps 3 --warmup 1 > gpurun_out/train_dp_swin.log 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/train_dp_swin.log
echo "== train-dp res B=4";  timeout 400 python bench.py --mode train-dp --variant res --batch 4 --steps 3 --warmup 1 > gpurun_out/train_dp_res.log 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/train_dp_res.log
grep -h "full_size\|\"swin\"\|hoist_ab" gpurun_out/parity_report.jsonl | head -30
