#!/usr/bin/env python3
"""Instruction mix of the LOOPS of one kernel in a hipcc -S listing (loops = backward branches):
    tools/loopmix.py file.s <substring of the mangled name> [min instructions]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    lines = [l.split(';')[0].strip() for l in body.split('\n')]
    lines = [l for l in lines if l and not l.startswith(';') and not (l.startswith('.') and not l.endswith(':'))]
    pos = {}
    for i, l in enumerate(lines):
        if l.endswith(':') or re.match(r'^\.?\w+:', l):
            pos[l.split(':')[0]] = i
    print(name, len(lines))
    for i, l in enumerate(lines):
        mm = re.match(r's_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)', l)
        if mm:
            t = mm.group(1) or mm.group(2)
            if t in pos and pos[t] < i and i - pos[t] >= mn:
                ins = [x.split()[0] for x in lines[pos[t]:i + 1] if not x.endswith(':')]
                c = Counter(ins)
                cls = Counter()
                for k, v in c.items():
                    g = ('mfma' if 'mfma' in k else 'valu' if k.startswith('v_') else 'lds' if k.startswith('ds_') else
                         'vmem' if k.startswith(('global_', 'buffer_', 'flat_')) else 'salu' if k.startswith('s_') else 'other')
                    cls[g] += v
                print('  loop %s..%d: %d instr  %s' % (t, i, len(ins), dict(cls)))
                print('     ' + '  '.join('%s %d' % kv for kv in c.most_common(14)))
