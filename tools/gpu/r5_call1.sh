#!/bin/bash
# Round 5, call 1: the GPU suite with the round's new tests (f16r head goldens, T = 50, launcher paths, default profile), smoke, the default bench line in its new
# form (three timed regions, top-level named_dtype / abs_clean, rocprofv3 clock on top), and the A/B of the streaming conv4's XCD-aware tile map.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl gpurun_out/bench_*.log
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/pytest_gpu.log
echo "== smoke";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 6 gpurun_out/smoke.log
X="--no-train-extra --no-nlspn-extra --no-head-extra --no-cpu-baseline --no-latency-b1 --no-streams-extra"
line() { name=$1; shift; timeout 600 python bench.py "$@" > gpurun_out/bench_$name.log 2>&1; echo "== $name rc=$?"; tail -n 1 gpurun_out/bench_$name.log | cut -c1-300; }
for i in 1 2; do
  line xcd1_s1_$i --steps 10 --warmup 3 --streams 1 $X
  line xcd0_s1_$i --steps 10 --warmup 3 --streams 1 --set thin_xcd=0 $X
  line xcd1_s2_$i --steps 10 --warmup 3 $X
  line xcd0_s2_$i --steps 10 --warmup 3 --set thin_xcd=0 $X
done
line default --steps 20 --warmup 3
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_xcd*.log")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f, d["value"], d["roofline"]["per_layer_avg_us"], d["spread"]["timed_regions_maps_per_s"])
    except Exception as e:
        print(f, "unparsed", e)
PY
