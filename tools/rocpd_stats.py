#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (what `rocprofv3 --kernel-trace --stats` writes on ROCm 7.2)
into the per-kernel statistics table that is committed under profiles/.
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/history/r01_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(grid_y), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid (x,y) | wg | LDS B | vgpr | agpr | sgpr |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0]
        if len(name) > 110:
            name = name[:107] + "..."
        print(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | {100 * r[2] / total:.1f} "
              f"| {r[6]},{r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} | {r[12]} |")


if __name__ == "__main__":
    main(sys.argv[1])
