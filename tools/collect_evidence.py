#!/usr/bin/env python3
"""Copies what tools/gpu/r2_final.sh left under gpurun_out/ into profiles/ under a run tag (the last line of every bench log = its JSON line).
    python tools/collect_evidence.py r02_run25"""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
logs = ["bench_c3_bf16_b4", "bench_c3_bf16_b1", "bench_c3_bf16_b16", "bench_c3_f16_b4", "bench_c3_fp32_b4", "bench_c2_nyu_bf16_b4", "bench_swin_bf16_b4", "bench_swin_f16_b4",
        "bench_c5_swin_f16_t50_b1", "bench_launcher_n1", "train_dp_swin_b4", "train_dp_res_b4"]
for n in logs:
    src = os.path.join(G, n + ".log")
    if os.path.exists(src):
        last = [l for l in open(src).read().splitlines() if l.startswith("{")]
        if last:
            name = n + ("_n1" if n.startswith("train_dp") else "")
            open(os.path.join(P, f"{tag}_{name}.json"), "w").write(last[-1] + "\n")
            print("json", name)
for src, dst in (("pytest_gpu.log", "pytest_gpu.txt"), ("smoke.log", "smoke.txt"), ("parity_report.jsonl", "parity_report.jsonl"),
                 ("prof_bf16/bench_kernel_stats.csv", "kernel_stats_bf16_kitti_b4.csv"), ("pmc_summary.txt", "pmc_bf16_kitti_b4.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{tag}_{dst}")); print("copy", dst)
if os.path.exists(os.path.join(G, "pmc_traffic.json")):
    shutil.copy(os.path.join(G, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json")); print("copy pmc_traffic.json")
