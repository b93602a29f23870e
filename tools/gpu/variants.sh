#!/bin/bash
# Runs tools/variant_bench.py for every build_variants/libddepth_<name>.so named on the command line (default: all)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
names="$@"; [ -z "$names" ] && names=$(ls build_variants/libddepth_*.so | sed 's/.*libddepth_//; s/\.so//' | grep -v hip_prof)
for n in $names; do
  [ -n "$VARIANT_OPTS_B" ] && DD_OPTS="$VARIANT_OPTS_B" DDEPTH_LIBRARY=$PWD/build_variants/libddepth_$n.so timeout 300 python tools/variant_bench.py 4 1 2>&1 | grep "^\["
  DDEPTH_LIBRARY=$PWD/build_variants/libddepth_$n.so timeout 300 python tools/variant_bench.py 4 1 2>&1 | grep "^\["
done
