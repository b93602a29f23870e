#!/usr/bin/env python3
"""Basic blocks of one kernel in a gfx950 .s file (hipcc -S --cuda-device-only) with per-category instruction counts; prints the blocks that
contain MFMAs or close a loop (back edges).  `from isa_blocks import blocks, cat` gives the parsed form (used for the static loop budgets in
DESIGN.md section 7 / profiles/r01_wino_isa_budget.md).
    python tools/isa_blocks.py file.s <mangled kernel name>"""
import re, sys, collections
def cat(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('global_','buffer_','scratch_','flat_')): return 'vmem'
    if op == 's_barrier': return 'barrier'
    if op == 's_waitcnt': return 'wait'
    if op.startswith('s_cbranch') or op == 's_branch': return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'
def blocks(path, name):
    s = open(path).read()
    i = s.index(name + ':'); j = s.index('s_endpgm', i)
    out = []; cur = ['entry', [], []]
    for l in s[i:j].split('\n')[1:]:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.') and not re.match(r'^\.LBB\d+_\d+:', t): continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            out.append(cur); cur = [m.group(1), [], []]; continue
        op = t.split()[0]
        cur[1].append(op)
        if cat(op) == 'branch': cur[2].append(t.split()[-1])
    out.append(cur)
    return out
if __name__ == '__main__':
    bl = blocks(sys.argv[1], sys.argv[2])
    idx = {b[0]: k for k, b in enumerate(bl)}
    for k, b in enumerate(bl):
        c = collections.Counter(cat(o) for o in b[1])
        back = [t for t in b[2] if t in idx and idx[t] <= k]
        if c.get('mfma') or back:
            dsd = collections.Counter(o for o in b[1] if o.startswith('ds_'))
            print(k, b[0], dict(c), 'BACKEDGE->' + ','.join(back) if back else '', dict(dsd))
