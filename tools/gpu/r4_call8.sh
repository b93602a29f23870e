#!/bin/bash
# Round 4, call 8: the refined mode on the Swin denoiser; the whole GPU suite after the clean-up; Swin bench lines f16 / f16r
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 gpurun_out/pytest_gpu.log
grep -h "swin_f16r\|swin_full_size" gpurun_out/parity_report.jsonl
X="--no-train-extra --no-nlspn-extra --no-abs-extra --no-streams-extra --no-parity-gate"
for prec in f16 f16r; do
  echo "== bench swin $prec"; timeout 500 python bench.py --variant swin --precision $prec --steps 5 --warmup 2 $X > gpurun_out/bench_swin_$prec.log 2>&1; echo "rc=$?"; tail -n 1 gpurun_out/bench_swin_$prec.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['cpu_baseline']; r = d['roofline']
print(d['value'], 'maps/s', d['ms_per_step'], 'ms/step; step frac', r['step_frac_of_peak'], 'layers', r['per_layer_avg_us'])
print('  near rmse %.3e  far rmse %.3e (to %.1f m)' % (c['gpu_vs_cpu_depth_rmse'], c['far_range']['gpu_vs_cpu_depth_rmse'], c['far_range']['depth_range_m'][1]), c.get('rmse_gate'))
print('  b1', d.get('latency_b1'), 'head', d.get('head_forward'))"
done
