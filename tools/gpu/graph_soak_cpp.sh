#!/bin/bash
# Round 6 (profiles/r06_experiments.md section 10): the graph-replay soak OUTSIDE torch (tools/micro/graph_replay_soak.cpp, built into build_variants/), with the HIP runtime of
# /opt/rocm (7.2) and with the torch wheel's (7.0, LD_LIBRARY_PATH), the runtime's graph fast path on / off, graphs / eager.
cd "$(dirname "$0")/../.."
TLR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))" 2>/dev/null)
# the wheel ships libamdhip64.so without its soname link: a directory that resolves libamdhip64.so.7 (and what it loads) to the wheel's files
mkdir -p /tmp/torch_hip && for f in $TLR/*.so*; do ln -sf $f /tmp/torch_hip/$(basename $f); done && ln -sf $TLR/libamdhip64.so /tmp/torch_hip/libamdhip64.so.7 && ln -sf $TLR/libhsa-runtime64.so /tmp/torch_hip/libhsa-runtime64.so.1
TL=/tmp/torch_hip
B=build_variants/graph_replay_soak
for rep in 1 2 3; do
  echo "[rocm, fast path on, graphs]   $(timeout 200 $B 1500 1 2 4 2>&1 | tail -1)"
  echo "[torch-lib, fast path on, graphs] $(LD_LIBRARY_PATH=$TL timeout 200 $B 1500 1 2 4 2>&1 | tail -1)"
  echo "[torch-lib, fast path on, graphs, + add_noise + denoise_once] $(LD_LIBRARY_PATH=$TL timeout 300 $B 1500 1 2 4 1 2>&1 | tail -1)"
done
echo "[rocm, fast path OFF, graphs]  $(DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 200 $B 1500 1 2 4 2>&1 | tail -1)"
echo "[torch-lib, fast path OFF, graphs] $(LD_LIBRARY_PATH=$TL DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 200 $B 1500 1 2 4 2>&1 | tail -1)"
echo "[torch-lib, fast path on, eager] $(LD_LIBRARY_PATH=$TL timeout 200 $B 1500 0 2 4 2>&1 | tail -1)"
echo "[torch-lib, fast path on, graphs, one lane] $(LD_LIBRARY_PATH=$TL timeout 200 $B 1500 1 1 4 2>&1 | tail -1)"
