#!/usr/bin/env python3
"""Average rocprofv3 PMC counters per kernel name from counter_collection CSVs (one directory per pass)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    for tag in ("conv_igemm2_kernel", "conv_igemm_kernel"):
        if tag in name:
            i = name.index("Cfg")
            return tag.replace("_kernel", "") + "<" + name[i:i + 14].split(">")[0] + ">"
    return name.split("(")[0][-40:]


def main(dirs):
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print(d, ": no counter_collection.csv")
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(lambda: collections.defaultdict(int))
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row.get("Kernel_Name", "?"))
                    c = row.get("Counter_Name")
                    v = float(row.get("Counter_Value", 0) or 0)
                    acc[k][c] += v
                    cnt[k][c] += 1
        print("==", d)
        for k in sorted(acc, key=lambda k: -sum(cnt[k].values())):
            if "conv_igemm" not in k and "conv4_stream" not in k:
                continue
            parts = [f"{c}={acc[k][c] / max(cnt[k][c], 1):.4g}" for c in sorted(acc[k])]
            n = max(cnt[k].values())
            print(f"  {k} (n={n}): " + "  ".join(parts))


if __name__ == "__main__":
    main(sys.argv[1:])
