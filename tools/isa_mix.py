#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S device assembly (per basic block): tools/isa_mix.py file.s <mangled-substr>"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_ZN') and key in l and l.rstrip().split(':')[0].endswith(key) or (l.startswith('_ZN') and key in l.split(':')[0]))
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
blocks, cur, name = [], Counter(), 'entry'
for l in s[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith((';', '.')) and not t.endswith(':'):
        continue
    if t.endswith(':') or re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append((name, cur)); cur, name = Counter(), t.split(':')[0]
        continue
    op = t.split()[0]
    cur[op] += 1
blocks.append((name, cur))
tot = Counter()
for n, c in blocks:
    tot += c
    k = sum(c.values())
    if k >= 12:
        mf = sum(v for o, v in c.items() if 'mfma' in o)
        va = sum(v for o, v in c.items() if o.startswith('v_') and 'mfma' not in o)
        print(f"{n:14s} {k:6d} instr  mfma {mf:4d}  valu {va:5d}  ds {sum(v for o,v in c.items() if o.startswith('ds_')):4d}  vmem {sum(v for o,v in c.items() if o.startswith(('buffer_','global_'))):4d}  waitcnt {c['s_waitcnt']:4d} nop {c['s_nop']:4d} accrd {c['v_accvgpr_read_b32']:4d} accwr {c['v_accvgpr_write_b32']:4d} mov {c['v_mov_b32']:4d}")
print('TOTAL', sum(tot.values()))
for o, v in tot.most_common(30):
    print(f"  {v:6d} {o}")
