#!/usr/bin/env python3
"""HBM traffic per launch of the fused conv kernels from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
passes as MI355X_MICROARCH.md prescribes).  Units/corrections per that guide: the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced read -> doubled; WRITE_SIZE is taken as is.
    python tools/pmc_traffic.py gpurun_out/pmc_c gpurun_out/pmc_d "<bench args>" > profiles/pmc_traffic.json"""
import collections, csv, glob, json, os, re, sys


def per_kernel(d, counter):
    """mean counter value per kernel over the dispatches with the LARGEST grid (the timed batch; bench.py also runs a
    B=1 latency leg whose smaller dispatches must not be averaged in)"""
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row.get("Kernel_Name", "")
            m = re.search(r"conv_igemm2?_kernel<dd::Cfg2?<(\d+), (\d+)>", name)
            if m:
                # kernel ids 48 / 49 are layers 8 / 9 on 16x32 tiles, 50 / 52 / 53 the hoisted forms of the Swin layers 5 / 7 / 7 (53: pred.0 o convB as one 5x5 kernel) (dd_kernels.h): booked under the layer they implement
                # 46 = layer 9 on 8x32 tiles with one patch buffer; 51 = the 5x5 form on 16x32 tiles
                layer = {46: 9, 48: 8, 49: 9, 50: 5, 51: 7, 52: 7, 53: 7}.get(int(m.group(2)), int(m.group(2)))
                key = f"layer{layer}_ek{m.group(1)}"
            else:
                m = re.search(r"conv4_stream_kernel<(\d+)[,>]", name)      # dd_thin.hip: conv4 as the persistent streaming kernel (<kind, stacked, input kind, operand pair>)
                if not m:
                    continue
                key = f"layer4_ek{m.group(1)}"
            rows[key].append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    out = {}
    for k, v in rows.items():
        g = max(x[0] for x in v)
        sel = [x[1] for x in v if x[0] == g]
        out[k] = sum(sel) / len(sel)
    return out


def lib_source_sha():
    """the stamp bench.py compares (bench.lib_source_sha): sha256 over csrc + include"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    for d in (os.path.join(root, "diffusiondepth_amd", "csrc"), os.path.join(root, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".hip", ".cpp")):
                hsh.update(f.encode())
                hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


if __name__ == "__main__":
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    import time
    out = {"bench_args": sys.argv[3] if len(sys.argv) > 3 else "", "lib_source_sha": lib_source_sha(), "taken": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()),
           "note": "bytes per launch; fetch = FETCH_SIZE KiB * 1024 * 2 (gfx950 wide-read correction), write = WRITE_SIZE KiB * 1024",
           "kernels": {k: {"fetch_bytes": fetch[k] * 1024 * 2, "write_bytes": write.get(k, 0.0) * 1024,
                           "hbm_bytes": fetch[k] * 1024 * 2 + write.get(k, 0.0) * 1024} for k in sorted(fetch)}}
    print(json.dumps(out, indent=1))
