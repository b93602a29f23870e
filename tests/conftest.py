import json
import os
import sys

# A pytest-xdist worker of the CPU suite (pytest_cmdline_main below) keeps its BLAS / OpenMP pools small: the workers are the parallelism.
# (In front of the numpy / torch imports: the pools read these once.)
if os.environ.get("PYTEST_XDIST_WORKER"):
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    os.environ.setdefault("MKL_NUM_THREADS", "2")

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")      # before any HIP call of the session: what `import diffusiondepth_amd` exports too (profiles/r06_experiments.md section 10)

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


_CONTROLLER_MARK = "DDEPTH_XDIST_CONTROLLER"      # set by the controller below, inherited by everything it starts


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`-m "not gpu"` (the CPU suite: oracle vs goldens, host logic, the kernels in the single-threaded host emulation) spreads over the host's cores with
    pytest-xdist when that plugin is installed and the caller chose no `-n`: 12 minutes serial, about 3 on 8 cores.  Never for `-m gpu` (one GPU, timed
    tests, its own process-group tests), never with DDEPTH_TESTS_SERIAL set.  The shared host-emulation build is lock-protected
    (hostemu_util._BuildLock), every rendezvous port is taken from the OS, tmp paths are per test.

    xdist runs THIS hook again inside every worker, after resetting the worker's `numprocesses` to None -- so a worker must never be mistaken for
    "the caller chose no -n" (it would start workers of its own, and those theirs).  Three independent guards: xdist's worker environment variable
    (set before the worker parses its configuration), the worker's `workerinput`, and a mark this controller leaves in the environment its
    workers inherit."""
    if os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get(_CONTROLLER_MARK) or hasattr(config, "workerinput"):
        return None
    opt = config.option
    if getattr(opt, "numprocesses", "absent") is not None or os.environ.get("DDEPTH_TESTS_SERIAL"):
        return None                                                # no xdist, or an explicit -n
    if (getattr(opt, "markexpr", "") or "").replace(" ", "") != "notgpu" or getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    n = min(os.cpu_count() or 1, 8)
    if n > 1:
        os.environ[_CONTROLLER_MARK] = str(os.getpid())
        opt.numprocesses = n
        opt.dist = "load"
        opt.tx = ["popen"] * n
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def cases():
    with open(os.path.join(GOLDEN_DIR, "cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    """golden('loop_res') -> dict of arrays minted from the reference (tests/golden/make_golden.py)."""
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load
