#!/bin/bash
# kernel trace of tools/lanes_head_probe.py (head eval forward with two lanes): where does the erratic slow case spend its time?
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do python tools/lanes_head_probe.py 2 none 2>&1 | grep "loss noise"; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$OLDPWD/gpurun_out/prof_lanes" -o lanes --output-format csv -- python "$OLDPWD/tools/lanes_head_probe.py" 2 none > "$OLDPWD/gpurun_out/lanes_probe.log" 2>&1)
grep "loss noise\|inference only" gpurun_out/lanes_probe.log
for f in $(find gpurun_out/prof_lanes -name "*kernel_trace.csv" | head -1); do
  python - "$f" gpurun_out/lanes_trace.tsv.gz <<'PY'
import csv, gzip, sys
with open(sys.argv[1], newline="") as f, gzip.open(sys.argv[2], "wt") as g:
    for r in csv.DictReader(f):
        g.write("%s\t%s\t%s\t%s\t%s\n" % (r["Kernel_Name"][:70].replace("\t", " "), r["Start_Timestamp"], r["End_Timestamp"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
PY
  rm -f "$f"
done
