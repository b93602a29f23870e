#!/bin/bash
# The evidence run of a round (tools/gpu/evidence.sh <tag>, e.g. r06_final): full GPU suite, smoke, PMC passes + rocprofv3 kernel stats of the ONE-STREAM default line (stamped with the sources' hash: the
# default bench line reads both), then the bench lines of BASELINE configs 2..5 and the other precisions.  Everything lands under gpurun_out/ (copied to profiles/).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-evidence}
rm -f gpurun_out/parity_report.jsonl gpurun_out/bench_*.log gpurun_out/${TAG}_*
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 5 gpurun_out/pytest_gpu.log
echo "== smoke";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 4 gpurun_out/smoke.log
echo "== pmc (f16r, KITTI B=4, one stream)"; bash tools/gpu/pmc.sh > gpurun_out/pmc.log 2>&1; tail -n 3 gpurun_out/pmc.log; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
echo "== rocprof (one stream: per-kernel durations)"
CFG="--precision f16r --batch 4 --size kitti --variant res"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_f16r" -o bench --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --streams 1 $CFG --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra > "$OLDPWD/gpurun_out/rocprof_f16r.log" 2>&1); echo "rocprof rc=$?"
KS=$(find gpurun_out/prof_f16r -name "*kernel_stats.csv" | head -1); head -n 12 "$KS" | cut -c1-200
python tools/kernel_stats_stamp.py "$KS" "$CFG" > gpurun_out/kernel_stats.json && cp gpurun_out/kernel_stats.json profiles/kernel_stats.json
find gpurun_out/prof_f16r -name "*kernel_trace.csv" -delete
X="--no-train-extra --no-nlspn-extra --no-head-extra"
line() { name=$1; shift; timeout 600 python bench.py "$@" > gpurun_out/bench_$name.log 2>&1; echo "== $name rc=$?"; tail -n 1 gpurun_out/bench_$name.log | cut -c1-420; }
line c3_f16r_b4 --steps 20 --warmup 3
line c3_bf16_b4 --steps 10 --warmup 2 --precision bf16 $X --no-parity-gate
line c3_f16_b4 --steps 10 --warmup 2 --precision f16 $X --no-parity-gate
line c3_f16x3_b4 --steps 5 --warmup 2 --precision f16x3 $X
line c3_fp32_b4 --steps 3 --warmup 1 --precision fp32 $X --no-latency-b1
line c3_f16r_b1 --steps 20 --warmup 3 --batch 1 --no-cpu-baseline $X
line c3_f16r_b16 --steps 5 --warmup 2 --batch 16 --no-cpu-baseline $X --no-latency-b1
line c2_nyu_f16r_b4 --steps 20 --warmup 3 --size nyu --no-train-extra --no-nlspn-extra
line c2_nyu_f16r_b28 --steps 10 --warmup 2 --size nyu --batch 28 --no-cpu-baseline $X --no-latency-b1
line swin_f16r_b4 --steps 5 --warmup 2 --variant swin --no-train-extra --no-nlspn-extra
line swin_f16_b4 --steps 5 --warmup 2 --variant swin --precision f16 $X --no-parity-gate
line swin_bf16_b4 --steps 5 --warmup 2 --variant swin --precision bf16 --no-cpu-baseline $X
line swin_f16x3_b4 --steps 3 --warmup 1 --variant swin --precision f16x3 $X --no-latency-b1
line c5_swin_f16r_t50_b1 --variant swin --T 50 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra
line train_dp_swin_b4 --mode train-dp --variant swin --batch 4 --steps 3 --warmup 1
line train_dp_res_b4 --mode train-dp --variant res --batch 4 --steps 3 --warmup 1
line train_dp_swin_f16x3_b4 --mode train-dp --variant swin --precision f16x3 --batch 4 --steps 3 --warmup 1
line train_dp_res_f16x3_b4 --mode train-dp --variant res --precision f16x3 --batch 4 --steps 3 --warmup 1
echo "== N=1 under the launcher"; timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 > gpurun_out/bench_launcher_n1.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_launcher_n1.log | cut -c1-300
echo "== rocprof swin f16r (one stream)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_swin" -o bench --output-format csv -- python "$OLDPWD/bench.py" --variant swin --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-nlspn-extra --no-head-extra > "$OLDPWD/gpurun_out/rocprof_swin.log" 2>&1); echo "rocprof rc=$?"
for f in $(find gpurun_out/prof_swin -name "*kernel_stats.csv" | head -1); do head -n 10 "$f" | cut -c1-200; done
find gpurun_out/prof_swin -name "*kernel_trace.csv" -delete

# the artefacts under the round's tag (copied to profiles/ by hand afterwards)
cp gpurun_out/pytest_gpu.log gpurun_out/${TAG}_pytest_gpu.txt; cp gpurun_out/smoke.log gpurun_out/${TAG}_smoke.txt; cp gpurun_out/parity_report.jsonl gpurun_out/${TAG}_parity_report.jsonl 2>/dev/null
cp gpurun_out/pmc_summary.txt gpurun_out/${TAG}_pmc_f16r_kitti_b4.txt 2>/dev/null
for f in gpurun_out/bench_*.log; do n=$(basename $f .log); tail -n 1 $f > gpurun_out/${TAG}_${n}.json; done
KS=$(find gpurun_out/prof_f16r -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && cp "$KS" gpurun_out/${TAG}_kernel_stats_f16r_kitti_b4.csv
KS=$(find gpurun_out/prof_swin -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && cp "$KS" gpurun_out/${TAG}_kernel_stats_swin_f16r_kitti_b4.csv
ls gpurun_out/${TAG}_* | head -60
