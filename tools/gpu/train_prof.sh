#!/bin/bash
# rocprofv3 kernel statistics of one data-parallel training step (bench.py --mode train-dp): where the step's GPU time goes
# (library kernels vs the PyTorch-ROCm modules that run in .train() mode: FPN + codec with batch-statistics BatchNorm, losses, SGD).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
V=${1:-swin}; B=${2:-4}
(cd /tmp && timeout 800 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_train_$V" -o train --output-format csv -- python "$OLDPWD/bench.py" --mode train-dp --variant $V --batch $B --steps 3 --warmup 1 > "$OLDPWD/gpurun_out/rocprof_train_$V.log" 2>&1); echo "rocprof rc=$?"
tail -n 1 gpurun_out/rocprof_train_$V.log | cut -c1-600
for f in $(find gpurun_out/prof_train_$V -name "*kernel_stats.csv" | head -1); do head -n 45 "$f" | cut -c1-220; cp "$f" gpurun_out/train_${V}_b${B}_kernel_stats.csv; done
for f in $(find gpurun_out/prof_train_$V -name "*kernel_trace.csv" | head -1); do
  python - "$f" "gpurun_out/train_${V}_b${B}_kernel_trace.csv.gz" <<'PY'
import csv, gzip, sys
with open(sys.argv[1], newline="") as f, gzip.open(sys.argv[2], "wt") as g:
    for r in csv.DictReader(f):
        g.write("%s\t%s\t%s\n" % (r["Kernel_Name"][:90].replace("\t", " "), r["Start_Timestamp"], r["End_Timestamp"]))
PY
  rm -f "$f"
done
