#!/bin/bash
# A/B of library build variants (build_variants/lib_<name>.so): bench lines only, parity on the first
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in build_variants/lib_*.so; do
  name=$(basename $lib .so); export DDEPTH_LIBRARY=$PWD/$lib
  if [ "$PARITY" = "1" ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "v2 and (golden or oracle)" 2>&1 | tail -n 2; fi
  for cfg in ${CFGS:-bf16,1,res bf16,4,res}; do IFS=, read -r a b c d <<< "$cfg"
    timeout 300 python bench.py --steps 8 --warmup 2 --precision $a --batch $b --variant $c $d --no-cpu-baseline > gpurun_out/bench_v.log 2>&1
    python - "$name" "$a" "$b" "$c" "$d" <<'PY'
import json,sys
n,a,b,c=sys.argv[1:5]
f="gpurun_out/bench_v.log"
try:
    d=json.loads([x for x in open(f) if x.startswith("{")][-1]); r=d["roofline"]
    print(n, c, a, "B="+b, " ".join(sys.argv[5:]), d["value"], "maps/s loop_ms", r["loop_ms_graph"], "layers_us", r["per_layer_avg_us"])
except Exception as e: print(n, "bench parse failed", e); print(open(f).read()[-800:])
PY
  done
done
