#!/usr/bin/env python3
"""Instruction mix of the large basic blocks of one kernel in a hipcc -S listing: isa_mix.py file.s substring [min_block_size]."""
import re, sys
from collections import Counter
text = open(sys.argv[1]).read()
want = sys.argv[2]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    print(name)
    blocks, cur = {}, 'entry'
    for l in body.split('\n'):
        l = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', l):
            cur = l.split(':')[0]
            continue
        if l and not l.startswith(('.', ';')):
            blocks.setdefault(cur, []).append(l.split()[0])
    for b, ins in blocks.items():
        if len(ins) < minb:
            continue
        c = Counter()
        for i in ins:
            k = ('mfma' if i.startswith('v_mfma') else 'trans' if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', i) else 'valu' if i.startswith('v_')
                 else 'wait' if i.startswith('s_waitcnt') else 'salu' if i.startswith('s_')
                 else 'vmem' if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'lds' if i.startswith('ds_') else 'other')
            c[k] += 1
        print(' ', b, len(ins), dict(c))
