#!/usr/bin/env python3
"""Stages the REFERENCE's own Python classes of the hot path as bytecode under oracle/_ref/py/ (test infrastructure only).

    python oracle/ref_py/build_ref.py [--force]

Why: BASELINE.md section 3 plans the `cpu_baseline` of bench.py as the reference's own classes timed on the bench host, but /root/reference does
not exist on the GPU box.  The DCN leg solved that by compiling the reference's device code into oracle/_ref/libref_dcn.so; this is the same
move for a Python reference: every module the import shim (tests/golden/ref_import.py, make_golden_hahi.py) pulls from /root/reference/src --
scheduling_ddim.py, ddim_depth_estimate_res.py, ddim_depth_estimate_res_swin_add.py, mmbev_base_depth_refine.py, depth_transform.py, common.py, the
package __init__ files they import through and (round 6, for tests/test_reference_facade.py) the model facade diffusion_dcbase_model.py, the HAHI neck
hahi.py and the HAHI / Vis head modules -- is COMPILED where it lies (py_compile, unchecked-hash .pyc, no source text) into the mirrored path
under oracle/_ref/py/.  oracle/_ref/ is git-ignored (nothing of the reference enters the history) and travels to the GPU box with the
snapshot like the built .so files; there the shim imports the sourceless modules, so what bench.py times as `cpu_baseline.kind =
"reference"` is the reference's code object for code object.  Without /root/reference the staged tree is used as it is; build() returns
None when neither exists (bench.py then falls back to the port and says so).
"""
from __future__ import annotations

import json
import os
import py_compile
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_ref", "py")
STAMP = os.path.join(OUT_DIR, "STAGED.json")
REF_ROOT = os.environ.get("DD_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")
SHIM_DIR = os.path.join(ROOT, "tests", "golden")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "model", "head"))


def staged() -> bool:
    return os.path.isfile(STAMP)


def _modules_the_shim_loads():
    """{module name: source file} of everything ref_import.load_reference() imports from the reference tree (in a child interpreter)."""
    code = ("import json, os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); os.environ['DD_REFERENCE_ROOT'] = %r; import ref_import; ref_import.load_reference(); ref_import.load_reference_facade(); import make_golden_hahi; make_golden_hahi.load_hahi_reference(); "
            "src = os.path.realpath(%r) + os.sep; "
            "print(json.dumps({n: m.__file__ for n, m in sys.modules.items() if getattr(m, '__file__', None) and os.path.realpath(m.__file__).startswith(src)}))"
            % (SHIM_DIR, ROOT, REF_ROOT, REF_SRC))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("importing the reference through tests/golden/ref_import.py failed:\n" + r.stderr[-3000:])
    return json.loads(r.stdout.strip().split("\n")[-1])


def build(force: bool = False, verbose: bool = True):
    if not reference_available():
        return OUT_DIR if staged() else None
    if staged() and not force:
        try:
            st = json.load(open(STAMP))
            if all(os.path.exists(os.path.join(REF_SRC, rel)) and os.path.getmtime(os.path.join(REF_SRC, rel)) <= st["mtime"] for rel in st["files"]) \
                    and st.get("python") == list(sys.version_info[:2]):
                return OUT_DIR
        except Exception:  # noqa: BLE001
            pass
    mods = _modules_the_shim_loads()
    files = sorted({os.path.relpath(os.path.realpath(f), os.path.realpath(REF_SRC)) for f in mods.values()})
    os.makedirs(OUT_DIR, exist_ok=True)
    for rel in files:
        dst = os.path.join(OUT_DIR, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(os.path.join(REF_SRC, rel), cfile=dst, dfile="<reference>/src/" + rel, doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        if verbose:
            print("[oracle.ref_py] compiled", rel, flush=True)
    json.dump({"files": files, "mtime": max(os.path.getmtime(os.path.join(REF_SRC, rel)) for rel in files), "python": list(sys.version_info[:2]),
               "what": "bytecode of the reference modules tests/golden/ref_import.py loads; generated, git-ignored, do not commit"}, open(STAMP, "w"), indent=1)
    return OUT_DIR


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
