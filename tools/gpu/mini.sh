#!/bin/bash
# minimal iteration: v2 parity tests + bench lines (bf16 B=1, B=4)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "v2" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/pytest_gpu.log
for cfg in bf16,1,res bf16,4,res ${EXTRA_CFG}; do IFS=, read -r a b c <<< "$cfg"; set -- $a $b $c
  timeout 300 python bench.py --steps 10 --warmup 3 --precision $1 --batch $2 --variant $3 --no-cpu-baseline > gpurun_out/bench_$1_b$2_$3.log 2>&1
  python - "$1" "$2" "$3" <<'PY'
import json,sys
f=f"gpurun_out/bench_{sys.argv[1]}_b{sys.argv[2]}_{sys.argv[3]}.log"
try:
    d=json.loads([x for x in open(f) if x.startswith("{")][-1]); r=d["roofline"]
    print(sys.argv[3], sys.argv[1], "B="+sys.argv[2], d["value"], "maps/s  loop_ms", r["loop_ms_graph"], "loop_frac", r["loop_frac_of_peak"], "layers_us", r["per_layer_avg_us"])
except Exception as e: print("bench parse failed", e); print(open(f).read()[-1500:])
PY
done
