#!/usr/bin/env python3
"""Builds a variant of the library with extra compile flags into build_variants/libddepth_<name>.so (git-ignored; travels to the GPU box).
Only dd_igemm2.hip (and the dd_api*.cpp units when --api is given) are recompiled; the other objects come from the default build.

    python tools/build_variant.py c3_2 -DDD_C3=2
Use with DDEPTH_LIBRARY=build_variants/libddepth_<name>.so (tools/variant_bench.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from diffusiondepth_amd import build as b


def main():
    name, flags = sys.argv[1], [a for a in sys.argv[2:] if a != "--api"]
    b.build()                                             # default objects up to date
    outdir = os.path.join(ROOT, "build_variants"); os.makedirs(outdir, exist_ok=True)
    hipcc = b.find_hipcc()
    redo = ["dd_igemm2.hip", "dd_thin.hip"] + (["dd_api.cpp", "dd_api_weights.cpp", "dd_api_plans.cpp", "dd_api_train.cpp"] if "--api" in sys.argv else [])
    objs = []
    for src in b.SOURCES:
        base = os.path.splitext(src)[0]
        if src in redo:
            obj = os.path.join(outdir, f"{name}_{base}.o")
            cmd = [hipcc] + [f for f in b.FLAGS if f != "-shared"] + flags + ["-x", "hip", "-c", os.path.join(b.CSRC, src), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit(r.stdout + r.stderr)
            if r.stderr.strip():
                print(r.stderr[-3000:])
        else:
            obj = os.path.join(b.HERE, "_build", base + ".o")
        objs.append(obj)
    so = os.path.join(outdir, f"libddepth_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print(so)


if __name__ == "__main__":
    main()
