#!/usr/bin/env python3
"""rocprofv3 `--kernel-trace --stats` summary (…kernel_stats.csv) of a ONE-STREAM bench.py run -> profiles/kernel_stats.json, stamped with the library
sources' hash and the bench arguments, so that bench.py can print the dominant kernel's roofline fraction from the SAME clock source as the committed
profile (the line's live-hipEvent figure read 4.6 % kinder than the profile's) -- and only for the sources it was taken on.
    python tools/kernel_stats_stamp.py <kernel_stats.csv> "<bench args>" > profiles/kernel_stats.json"""
import csv, json, os, re, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_traffic import lib_source_sha  # noqa: E402  (the same stamp as profiles/pmc_traffic.json)

LAYER_OF = {46: 9, 48: 8, 49: 9, 50: 5, 51: 7, 52: 7, 53: 7}


def main():
    kernels = {}
    for row in csv.DictReader(open(sys.argv[1])):
        name = row["Name"]
        m = re.search(r"conv_igemm2_kernel<dd::Cfg2<(\d+), (\d+)>", name)
        if m:
            lid = int(m.group(2))
            key = f"layer{LAYER_OF.get(lid, lid)}_ek{m.group(1)}_id{lid}"
        else:
            m = re.search(r"conv4_stream_kernel<(\d+)[,>]", name)
            if not m:
                continue
            key = f"layer4_ek{m.group(1)}_stream"
        kernels[key] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "total_us": float(row["TotalDurationNs"]) / 1e3, "name": name}
    json.dump({"bench_args": sys.argv[2] if len(sys.argv) > 2 else "", "lib_source_sha": lib_source_sha(), "taken": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()),
               "note": "rocprofv3 --kernel-trace --stats of `bench.py --streams 1 ...` (one stream: a launch's duration is a property of the kernel)", "kernels": kernels},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
