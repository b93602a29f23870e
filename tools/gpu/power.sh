#!/bin/bash
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp
smp() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed -E 's/.*\((.*)Mhz\).*/sclk \1/; s/.*\(W\): (.*)/W \1/' | tr '\n' ' '; echo; }
for args in "0 4 0 8" "0 4 1 8" "1 4 0 8" "2 4 0 8" "1 4 0 4"; do
  ./build_variants/mfma_power $args &
  P=$!; sleep 2; smp; sleep 1; smp; wait $P
done
