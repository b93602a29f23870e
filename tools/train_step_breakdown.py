#!/usr/bin/env python3
"""One steady-state training iteration out of a rocprofv3 kernel trace of `bench.py --mode train-dp` (tools/gpu/train_prof.sh writes the
compact trace: kernel name, start, end per line, gzip).  The whole-run `--stats` table is useless for this question: MIOpen's find pass
(naive / every-solver benchmark launches at the first call of each convolution shape) dominates it.  Iterations are delimited by the first
BatchNorm-training kernel of each forward (the FPN of the head runs first); the last complete one is summarised.
    python tools/train_step_breakdown.py gpurun_out/train_swin_b4_kernel_trace.csv.gz [top]"""
import collections, gzip, sys

rows = sorted(((n, int(s), int(e)) for n, s, e in (l.rstrip("\n").split("\t") for l in gzip.open(sys.argv[1], "rt"))), key=lambda r: r[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
bn = [r[1] for r in rows if "BatchNormFwdTrain" in r[0]]
starts = [bn[0]] + [b for a, b in zip(bn, bn[1:]) if b - a > 60e6]
s, e = starts[-2], starts[-1]
sel = [r for r in rows if s <= r[1] < e]
busy = sum(r[2] - r[1] for r in sel)
lib = sum(r[2] - r[1] for r in sel if "dd::" in r[0])
print(f"{len(starts)} iterations in the trace; the last complete one: wall {(e - s) / 1e6:.1f} ms, kernels {busy / 1e6:.1f} ms = library (dd::) "
      f"{lib / 1e6:.1f} ms ({100 * lib / busy:.1f} %) + PyTorch-ROCm / MIOpen {(busy - lib) / 1e6:.1f} ms ({100 * (busy - lib) / busy:.1f} %)")
acc = collections.defaultdict(lambda: [0, 0])
for n, a, b in sel:
    acc[n[:80]][0] += 1; acc[n[:80]][1] += b - a
print("| ms | launches | kernel |\n|---|---|---|")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"| {t / 1e6:.2f} | {n} | `{k}` |")
