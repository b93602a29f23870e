#!/usr/bin/env python3
"""Instruction-category histogram of the kernels in a gfx950 .s file (hipcc -S --cuda-device-only).
    python tools/isa_hist.py file.s [name-substring ...]"""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    filt = sys.argv[2:]
    for m in re.finditer(r'^(_Z\w+):\s*; @', s, re.M):
        name = m.group(1)
        if filt and not any(f in name for f in filt):
            continue
        j = s.index('s_endpgm', m.end())
        ops = collections.Counter()
        for line in s[m.end():j].split('\n'):
            line = line.strip()
            if not line or line[0] in ';.' or line.endswith(':'):
                continue
            ops[line.split()[0]] += 1
        cats = collections.Counter()
        for op, c in ops.items():
            k = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'ds' if op.startswith('ds_')
                 else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'salu' if op.startswith('s_') else 'other')
            cats[k] += c
        print(name[:70], 'total', sum(ops.values()), dict(cats))
        print('   ', ops.most_common(24))


if __name__ == "__main__":
    main()
