#!/bin/bash
# Round 3 evidence run: full GPU suite, smoke, the bench lines of BASELINE configs 2..5 (+ the abs-clean f16x3 line), rocprofv3 kernel stats of
# the one-stream bench command, PMC passes (HBM traffic + instruction mix).  Everything lands under gpurun_out/ and is copied into profiles/.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 5 gpurun_out/pytest_gpu.log
echo "== smoke";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 5 gpurun_out/smoke.log
# PMC passes first: the default bench line below reads the traffic file stamped with THESE sources
echo "== pmc"; bash tools/gpu/pmc.sh > gpurun_out/pmc.log 2>&1; tail -n 3 gpurun_out/pmc.log; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
echo "== bench C3 default";   timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_c3_bf16_b4.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_c3_bf16_b4.log | cut -c1-1200
echo "== bench C3 f16x3 (abs-clean mode)"; timeout 300 python bench.py --steps 5 --warmup 2 --precision f16x3 --no-train-extra --no-nlspn-extra --no-head-extra > gpurun_out/bench_c3_f16x3_b4.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_c3_f16x3_b4.log | cut -c1-400
echo "== bench C3 B=1";       timeout 300 python bench.py --steps 20 --warmup 3 --batch 1 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra > gpurun_out/bench_c3_bf16_b1.log 2>&1; tail -n 1 gpurun_out/bench_c3_bf16_b1.log | cut -c1-400
echo "== bench C3 B=16";      timeout 300 python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 > gpurun_out/bench_c3_bf16_b16.log 2>&1; tail -n 1 gpurun_out/bench_c3_bf16_b16.log | cut -c1-400
echo "== bench C3 f16";       timeout 300 python bench.py --steps 10 --warmup 2 --precision f16 --no-train-extra --no-nlspn-extra --no-head-extra > gpurun_out/bench_c3_f16_b4.log 2>&1; tail -n 1 gpurun_out/bench_c3_f16_b4.log | cut -c1-400
echo "== bench C3 fp32";      timeout 300 python bench.py --steps 3 --warmup 1 --precision fp32 --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 > gpurun_out/bench_c3_fp32_b4.log 2>&1; tail -n 1 gpurun_out/bench_c3_fp32_b4.log | cut -c1-400
echo "== bench C2 NYU";       timeout 300 python bench.py --steps 20 --warmup 3 --size nyu --no-train-extra --no-nlspn-extra > gpurun_out/bench_c2_nyu_bf16_b4.log 2>&1; tail -n 1 gpurun_out/bench_c2_nyu_bf16_b4.log | cut -c1-400
echo "== bench swin bf16 T=20 (training precision)"; timeout 300 python bench.py --steps 5 --warmup 2 --variant swin --precision bf16 --no-cpu-baseline --no-train-extra --no-nlspn-extra > gpurun_out/bench_swin_bf16_b4.log 2>&1; tail -n 1 gpurun_out/bench_swin_bf16_b4.log | cut -c1-400
echo "== bench swin f16 T=20"; timeout 400 python bench.py --steps 5 --warmup 2 --variant swin --precision f16 --no-train-extra --no-nlspn-extra --no-head-extra > gpurun_out/bench_swin_f16_b4.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_swin_f16_b4.log | cut -c1-400
echo "== bench swin f16x3 T=20"; timeout 400 python bench.py --steps 3 --warmup 1 --variant swin --precision f16x3 --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 > gpurun_out/bench_swin_f16x3_b4.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_swin_f16x3_b4.log | cut -c1-400
echo "== C4 train-dp swin B=4"; timeout 400 python bench.py --mode train-dp --variant swin --batch 4 --steps 3 --warmup 1 > gpurun_out/train_dp_swin_b4.log 2>&1; tail -n 1 gpurun_out/train_dp_swin_b4.log | cut -c1-400
echo "== C4 train-dp res B=4"; timeout 400 python bench.py --mode train-dp --variant res --batch 4 --steps 3 --warmup 1 > gpurun_out/train_dp_res_b4.log 2>&1; tail -n 1 gpurun_out/train_dp_res_b4.log | cut -c1-400
echo "== N=1 under the launcher"; timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 > gpurun_out/bench_launcher_n1.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_launcher_n1.log | cut -c1-300
echo "== C5 swin f16 T=50 B=1 (+NLSPN extra)"; timeout 600 python bench.py --variant swin --precision f16 --T 50 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra > gpurun_out/bench_c5_swin_f16_t50_b1.log 2>&1; tail -n 1 gpurun_out/bench_c5_swin_f16_t50_b1.log | cut -c1-400
echo "== rocprof (one stream: per-kernel durations)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_bf16" -o bench --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra > "$OLDPWD/gpurun_out/rocprof_bf16.log" 2>&1); echo "rocprof rc=$?"
for f in $(find gpurun_out/prof_bf16 -name "*kernel_stats.csv" | head -1); do head -n 12 "$f" | cut -c1-200; done
find gpurun_out/prof_bf16 -name "*kernel_trace.csv" -delete
echo "== rocprof swin f16 (one stream)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_swin" -o bench --output-format csv -- python "$OLDPWD/bench.py" --variant swin --precision f16 --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra > "$OLDPWD/gpurun_out/rocprof_swin.log" 2>&1); echo "rocprof rc=$?"
for f in $(find gpurun_out/prof_swin -name "*kernel_stats.csv" | head -1); do head -n 10 "$f" | cut -c1-200; done
find gpurun_out/prof_swin -name "*kernel_trace.csv" -delete
