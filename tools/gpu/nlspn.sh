#!/bin/bash
# NLSPN / DCNv2: GPU parity tests + timing (short: ~1 min).  VARIANTS="g4,g1p,l16" also times / tests the kernel variants.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 200 python -m pytest tests/test_zz_gpu_nlspn.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_nlspn.log 2>&1; echo "pytest rc=$?"; tail -n 30 gpurun_out/pytest_nlspn.log
for v in ${TEST_VARIANTS}; do DD_NLSPN_KERNEL=$v timeout 100 python -m pytest tests/test_zz_gpu_nlspn.py -m gpu -q -p no:cacheprovider -k nlspn > gpurun_out/pytest_nlspn_$v.log 2>&1; echo "variant $v pytest rc=$?"; tail -n 3 gpurun_out/pytest_nlspn_$v.log; done
for b in 1 4; do
timeout 100 python tools/nlspn_timing.py --batch $b --variants "${VARIANTS}" > gpurun_out/nlspn_timing_b$b.json 2> gpurun_out/nlspn_timing_b$b.err; python - $b <<'PY'
import json,sys
try:
    d=json.load(open(f"gpurun_out/nlspn_timing_b{sys.argv[1]}.json")); v=d.pop("variants"); print({k:(round(x,4) if isinstance(x,float) else x) for k,x in d.items()})
    for k,x in v.items(): print("   ",k,{a:round(b,3) if b>1e-3 else b for a,b in x.items()})
except Exception as e: print("timing failed", e); print(open(f"gpurun_out/nlspn_timing_b{sys.argv[1]}.err").read()[-1500:])
PY
done
