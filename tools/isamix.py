#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -S listing: tools/isamix.py file.s <substring of the mangled name>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    lines = [l.strip() for l in body.split('\n')]
    ins = [l.split()[0] for l in lines if l and not l.startswith(('.', ';')) and not l.endswith(':')]
    c = Counter(ins)
    print(name, len(ins))
    for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
        print('   %-28s %d' % (k, v))
