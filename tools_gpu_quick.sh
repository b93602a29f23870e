#!/bin/bash
# quick iteration: parity tests (v2 only unless FULL=1), bench lines, PMC passes a+b
cd "$(dirname "$0")"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
KSEL="v2"; [ "$FULL" = "1" ] && KSEL="v1 or v2"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$KSEL" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 6 gpurun_out/pytest_gpu.log
for cfg in "bf16 1" "bf16 4" "f16 1"; do set -- $cfg
  timeout 300 python bench.py --steps 10 --warmup 2 --precision $1 --batch $2 --no-cpu-baseline > gpurun_out/bench_$1_b$2.log 2>&1
  python - "$1" "$2" <<'PY'
import json,sys
try:
    l=[x for x in open(f"gpurun_out/bench_{sys.argv[1]}_b{sys.argv[2]}.log") if x.startswith("{")][-1]; d=json.loads(l)
    r=d["roofline"]; print(sys.argv[1], "B="+sys.argv[2], d["value"], "maps/s  loop_ms", r["loop_ms_graph"], "loop_frac", r["loop_frac_of_peak"], "layers_us", r["per_layer_avg_us"])
except Exception as e: print("bench parse failed", e); print(open(f"gpurun_out/bench_{sys.argv[1]}_b{sys.argv[2]}.log").read()[-2000:])
PY
done
R="$PWD"
pass() { name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$R/gpurun_out/pmc_$name" -o p --output-format csv -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-graph > "$R/gpurun_out/pmc_$name.log" 2>&1)
  find gpurun_out/pmc_$name -name "*kernel_trace.csv" -delete; }
rm -rf gpurun_out/pmc_*
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > gpurun_out/pmc_summary.txt 2>&1; cat gpurun_out/pmc_summary.txt
