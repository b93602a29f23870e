#!/bin/bash
# Round 4, call 9: conv2 with its input patch staged in 16-channel chunks under the first split's MFMAs (-DDD_C2_CK16=1) against the default library
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
export DD_PRECS=bf16,f16r
for lib in diffusiondepth_amd/libddepth_hip.so build_variants/libddepth_c2k16.so diffusiondepth_amd/libddepth_hip.so build_variants/libddepth_c2k16.so; do
  echo "== $lib"
  DDEPTH_LIBRARY=$PWD/$lib timeout 400 python tools/variant_bench.py 4 1 2>&1 | grep "^\["
done
echo "== nlspn B=1 latency (graph cache)"
timeout 200 python - <<'PY'
import torch, bench
dev = torch.device("cuda", 0)
print({k: v for k, v in bench.nlspn_extra(dev, 4, 352, 1216).items() if k in ("module_forward_ms", "latency_b1_ms")})
import os
os.environ["DD_NLSPN_GRAPH"] = "0"
print("graph off:", {k: v for k, v in bench.nlspn_extra(dev, 4, 352, 1216).items() if k in ("module_forward_ms", "latency_b1_ms")})
PY
