#!/bin/bash
# Round-end verification in one short gpurun call: full GPU test suite -> smoke -> default bench line -> rocprofv3 kernel stats of the
# bench command -> NLSPN timing incl. the pipelined variants.  Logs under gpurun_out/.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest"; timeout 240 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 12 gpurun_out/pytest_gpu.log
DD_NLSPN_KERNEL=l16pq timeout 60 python -m pytest tests/test_zz_gpu_nlspn.py -m gpu -q -p no:cacheprovider -k nlspn > gpurun_out/pytest_nlspn_l16pq.log 2>&1; echo "variant l16pq pytest rc=$?"; tail -n 2 gpurun_out/pytest_nlspn_l16pq.log
echo "== smoke";  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 4 gpurun_out/smoke.log
echo "== bench";  timeout 240 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_bf16.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench_bf16.log
echo "== nlspn timing"
for b in 1 4; do timeout 60 python tools/nlspn_timing.py --batch $b --variants "l16p,l16pq,l8pq,l32pq" > gpurun_out/nlspn_timing_b$b.json 2> gpurun_out/nlspn_timing_b$b.err; python - $b <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/nlspn_timing_b{sys.argv[1]}.json")); v=d.pop("variants"); print({k:(round(x,4) if isinstance(x,float) else x) for k,x in d.items()})
    for k,x in v.items(): print("   ",k,{a:round(b,3) if b>1e-3 else b for a,b in x.items()})
except Exception as e: print("timing failed", e); print(open(f"gpurun_out/nlspn_timing_b{sys.argv[1]}.err").read()[-1500:])
PY
done
echo "== rocprof"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_final" -o bench --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-train-extra --no-latency-b1 > "$OLDPWD/gpurun_out/rocprof_final.log" 2>&1); echo "rocprof rc=$?"
for f in $(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1); do head -n 16 "$f"; done
find gpurun_out/prof_final -name "*kernel_trace.csv" -size +20M -delete
