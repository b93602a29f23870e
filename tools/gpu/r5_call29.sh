#!/bin/bash
# (record of a finished experiment: variants built with tools/micro/r05_nontemporal.patch applied: `python tools/build_variant.py ntst -DDD_NT_STORE=1`, `... ntld -DDD_NT_LOAD=1`, `... ntboth` with both; r5base = the unpatched build)
# Round 5, call 29: cache policy of the activation traffic of the large convolutions -- 16-byte epilogue stores as non-temporal stores (DD_NT_STORE), raw patch loads
# as non-temporal loads (DD_NT_LOAD), both; against the round's baseline library, A/B twice on one box.  Bit-identical results (same arithmetic).
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp DD_PRECS=f16r
run() { echo "== $1"; DDEPTH_LIBRARY=build_variants/libddepth_$1.so timeout 300 python tools/variant_bench.py 4 2>&1 | grep -v "amdgpu.ids" | tail -n 2; }
{
for i in 1 2; do run r5base; run ntst; run ntld; run ntboth; done
run r5base
} > gpurun_out/call29_nt.txt 2>&1
cat gpurun_out/call29_nt.txt
