#!/bin/bash
# N processes per arm of tools/nan_arms.py, interleaved (the 16-bit training step with its trajectory-keeping forward as a replayed hipGraph: profiles/r06_experiments.md section 5).
#   tools/gpu/soak.sh <out-name> <processes per arm> <arm> [<arm> ...]     arm = "[ENV=VALUE ...] <nan_arms arm> <train_graphs 0|1>"
#   e.g.  gpurun -- 'bash tools/gpu/soak.sh r06_arms 12 "head 1" "lib 1" "head_nomiopen 1" "GRAPH_FENCE=1 head 1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 head 1"'
# Output: gpurun_out/<out-name>.txt -- one line per process, then "failing of N" per arm.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/$1.txt; n=$2; shift 2
: > "$out"
for i in $(seq 1 "$n"); do
  k=0
  for arm in "$@"; do
    k=$((k + 1))
    line=$(env $(echo "$arm" | tr ' ' '\n' | grep '=' | tr '\n' ' ') timeout 300 python tools/nan_arms.py $(echo "$arm" | tr ' ' '\n' | grep -v '=' | tr '\n' ' ') 40 2>&1 | grep "^\[" | tail -n 1)
    echo "arm$k $line" >> "$out"
  done
done
k=0
for arm in "$@"; do k=$((k + 1)); echo "arm$k ($arm): $(grep "^arm$k " "$out" | grep -vc 'bad iterations: 0') failing of $(grep -c "^arm$k " "$out")"; done | tee -a "$out"
