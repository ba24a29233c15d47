#!/usr/bin/env python3
"""Instruction mix of the big basic blocks of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: tools/asm_blocks.py <file.s> <kernel-name-substring> [min-instructions]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(pat) + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
blocks, cur = [], ["entry", []]
blocks.append(cur)
for l in lines[start + 1:end]:
    l = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), []]; blocks.append(cur)
    elif l and not l.startswith((';', '.')):
        cur[1].append(l.split(';')[0].strip())
for name, ins in blocks:
    if len(ins) >= mn:
        c = Counter(i.split()[0] for i in ins)
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        f64 = sum(v for k, v in c.items() if 'f64' in k)
        br = [i for i in ins if i.startswith(('s_cbranch', 's_branch'))]
        print(f"{name}: {len(ins)} instr, VALU {valu}, f64 {f64}, branches {len(br)}")
        print("   ", dict(c.most_common(18)))
