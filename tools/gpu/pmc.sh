#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over a short eager bf16 bench; summaries land in gpurun_out/pmc_*.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
CFG="${BENCH_CFG:---precision f16r --batch 4 --size kitti --variant res}"
ARGS="--steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-train-extra --no-latency-b1 --no-streams-extra --no-head-extra --no-nlspn-extra --no-graph $CFG"
(cd /tmp && rocprofv3 -L > "$R/gpurun_out/counters_list.txt" 2>&1)
grep -c . gpurun_out/counters_list.txt
pass() {  # name, counters...
  name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$R/gpurun_out/pmc_$name" -o p --output-format csv -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_$name.log" 2>&1)
  echo "pass $name rc=$?"
  find gpurun_out/pmc_$name -name "*kernel_trace.csv" -delete
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass c FETCH_SIZE GRBM_GUI_ACTIVE
pass d WRITE_SIZE GRBM_GUI_ACTIVE
pass e SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
ls -la gpurun_out/pmc_*/ | head -40
python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c gpurun_out/pmc_d gpurun_out/pmc_e > gpurun_out/pmc_summary.txt 2>&1
cat gpurun_out/pmc_summary.txt | head -60
python tools/pmc_traffic.py gpurun_out/pmc_c gpurun_out/pmc_d "$CFG" > gpurun_out/pmc_traffic.json; cat gpurun_out/pmc_traffic.json
