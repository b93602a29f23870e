#!/usr/bin/env python3
"""Static instruction mix of the convolution kernels' code objects: compiles dd_igemm2.hip to assembly (device only, no GPU needed) and prints, per
selected instantiation, the MFMA / VALU / LDS / VMEM instruction counts and the most frequent VALU opcodes (static counts: loops are counted once).
    python tools/isa_mix.py 'Cfg2<(2|5), (1|2|46)>' [extra hipcc flags]"""
import os, re, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else r"Cfg2<(2|5), (1|2|46)>"
src = os.path.join(ROOT, "diffusiondepth_amd", "csrc", "dd_igemm2.hip")
out = "/tmp/dd_igemm2_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-x", "hip",
                "--cuda-device-only", "-S", "-o", out, src] + sys.argv[2:], check=True, capture_output=True)
s = open(out).read()
parts = re.split(r"\n(_ZN2dd18conv_igemm2_kernel[^\n:]*):[^\n]*\n", s)
names, bodies = parts[1::2], parts[2::2]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for n, b in zip(dem, bodies):
    if not re.search(pat, n):
        continue
    b = b.split(".Lfunc_end")[0]
    ops = re.findall(r"^\s+([a-z_0-9]+)", b, flags=re.M)
    c = Counter(ops)
    tot = lambda f: sum(v for k, v in c.items() if f(k))
    print(n)
    print("   mfma", tot(lambda k: k.startswith("v_mfma")), "valu", tot(lambda k: k.startswith("v_") and not k.startswith("v_mfma")),
          "salu", tot(lambda k: k.startswith("s_")), "ds_read", tot(lambda k: k.startswith("ds_read")), "ds_write", tot(lambda k: k.startswith("ds_write")),
          "global_load", tot(lambda k: k.startswith("global_load")), "global_store", tot(lambda k: k.startswith("global_store")),
          "scratch", tot(lambda k: k.startswith("scratch_")))
    print("   ", [(k, v) for k, v in c.most_common(60) if k.startswith("v_") and not k.startswith("v_mfma")][:30])
