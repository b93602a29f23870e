#!/usr/bin/env python3
"""Copies what tools/gpu/r4_final.sh (rN_final.sh) left under gpurun_out/ into profiles/ under a run tag (the last JSON line of every bench log).
    python tools/collect_evidence.py r04_run9"""
import glob, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for src in sorted(glob.glob(os.path.join(G, "bench_*.log"))):
    n = os.path.basename(src)[:-4]
    last = [l for l in open(src).read().splitlines() if l.startswith("{")]
    if last:
        open(os.path.join(P, f"{tag}_{n}.json"), "w").write(last[-1] + "\n")
        print("json", n)
for src, dst in (("pytest_gpu.log", "pytest_gpu.txt"), ("smoke.log", "smoke.txt"), ("parity_report.jsonl", "parity_report.jsonl"),
                 ("prof_f16r/bench_kernel_stats.csv", "kernel_stats_f16r_kitti_b4.csv"), ("prof_swin/bench_kernel_stats.csv", "kernel_stats_swin_f16r_kitti_b4.csv"),
                 ("pmc_summary.txt", "pmc_f16r_kitti_b4.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{tag}_{dst}")); print("copy", dst)
for stamped in ("pmc_traffic.json", "kernel_stats.json"):
    if os.path.exists(os.path.join(G, stamped)):
        shutil.copy(os.path.join(G, stamped), os.path.join(P, stamped)); print("copy", stamped)
