"""CPU: the multi-rank plumbing of bench.py (`python bench.py --gpus N` silently measured ONE GPU).
`python bench.py --gpus 2` -- invoked plainly, as the driver does -- must start 2 ranks itself, and the JSON line's n_gpus must be the
world size the process group reports.  The hot path has no CPU fallback, so on this GPU-less host the ranks run --dist-selftest (same
spawn / rendezvous / barrier / MAX-over-ranks code, gloo backend); without it the command must fail cleanly, not fall back."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=600, env=e)


def test_plain_invocation_spawns_the_ranks_itself():
    r = _run("--gpus", "2", "--dist-selftest", "--steps", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["requested_gpus"] == 2 and d["backend"] == "gloo"
    assert d["units_summed"] == 3 * 1 + 3 * 2                           # the census the real N > 1 line uses (bench.rank_census): units summed over ranks
    assert d["ms_per_step"] >= 2.0                                      # MAX over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1 ms
    assert "starting 2 ranks" in r.stderr


def test_a_launcher_world_size_that_contradicts_gpus_is_refused():
    r = _run("--gpus", "4", "--dist-selftest", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_without_a_gpu_the_benchmark_fails_cleanly_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        return                                                          # GPU box: covered by the real bench run
    for argv in (("--gpus", "2"), ("--gpus", "1")):
        r = _run(*argv)
        assert r.returncode != 0
        assert "no CPU fallback" in r.stderr + r.stdout
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]     # and no JSON line pretending to be a measurement
