#!/bin/bash
# Round 6: the default line under torch.distributed.run with one rank is 23 % slower per step.  Which part of the launcher does it?  (the process group alone does not:
# tools/rccl_slowdown_probe.py)  Arms: plain; the process-group path without the launcher (WORLD_SIZE=1 in the environment); the same with OMP_NUM_THREADS=1; the launcher;
# the launcher's environment replayed without the launcher.
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
X="--steps 10 --warmup 2 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra --no-abs-extra"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print('$1', o['value'], o['spread']['step_ms_median'], o['spread']['timed_regions_maps_per_s'])
"; }
PG="WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
timeout 300 python bench.py $X 2>/dev/null | pick plain
env $PG MASTER_PORT=29561 timeout 300 python bench.py --gpus 1 $X 2>/dev/null | pick pg_without_launcher
env $PG MASTER_PORT=29562 OMP_NUM_THREADS=1 timeout 300 python bench.py --gpus 1 $X 2>/dev/null | pick pg_without_launcher_omp1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 1 $X 2>/dev/null | pick launcher
# the launcher's environment, dumped by a child and replayed
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29564 tools/gpu/dump_env.py > gpurun_out/launcher_env.json 2>/dev/null
python - <<'PY'
import json, os, subprocess, sys
child = json.load(open("gpurun_out/launcher_env.json"))
new = {k: v for k, v in child.items() if os.environ.get(k) != v}
print("variables the launcher adds / changes:", json.dumps({k: v[:60] for k, v in new.items()}))
X = "--steps 10 --warmup 2 --no-cpu-baseline --no-train-extra --no-nlspn-extra --no-head-extra --no-latency-b1 --no-streams-extra --no-abs-extra".split()
def run(tag, env):
    e = dict(os.environ); e.update(env); e["MASTER_PORT"] = "29570"
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + X, env=e, capture_output=True, text=True, timeout=300).stdout
    for l in out.splitlines():
        if l.startswith("{"):
            o = json.loads(l); print(tag, o["value"], o["spread"]["step_ms_median"], flush=True)
run("replayed_launcher_env", new)
base = {k: new[k] for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR") if k in new}
for k in new:
    if k in base or k == "MASTER_PORT":
        continue
    run("pg + " + k, {**base, k: new[k]})
PY
