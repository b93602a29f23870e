#!/usr/bin/env python3
"""Register / scratch / occupancy report of every kernel in diffusiondepth_amd/csrc (SURVEY.md 8d "kernel-resource report"): compiles each
.hip for gfx950 with -Rpass-analysis=kernel-resource-usage (device side only; needs no GPU) and prints one markdown table per file.
    python tools/kernel_resources.py > profiles/rNN_kernel_resources.md
Occupancy is the compiler's register-limited waves/SIMD; the LDS limit comes on top (dynamic LDS per workgroup: DESIGN.md section 3)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffusiondepth_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-x", "hip", "-c", "--cuda-device-only",
         "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull]
KEYS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except Exception:
        return names


def main():
    print("# Kernel resources (gfx950, hipcc -O3, `-Rpass-analysis=kernel-resource-usage`)\n")
    print("Register-limited occupancy as the compiler reports it; `LDS static` excludes the dynamic LDS the convolution kernels request at launch "
          "(DESIGN.md section 3 table).  Element kind in the template arguments: 0 = fp32, 1 = bf16, 2 = f16.\n")
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, f)], capture_output=True, text=True)
        rows, cur = [], None
        for line in r.stderr.split("\n"):
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z\[\]/ ]+?): (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
        if not rows:
            continue
        names = demangle([x["name"] for x in rows])
        print(f"## {f}\n")
        print("| kernel | VGPR | AGPR | SGPR | scratch B/lane | VGPR spill | SGPR spill | LDS static | waves/SIMD |")
        print("|---|---|---|---|---|---|---|---|---|")
        for x, n in zip(rows, names):
            n = re.sub(r"\(.*\)$", "", n).replace("dd::", "")
            print("| `" + n + "` | " + " | ".join(x.get(k, "?") for k in KEYS) + " |")
        print()


if __name__ == "__main__":
    sys.exit(main())
