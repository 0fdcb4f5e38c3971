"""Summarise a rocprofv3 --pmc run: per (kernel, grid size) in first-seen order, mean of every counter.
    python tools/pmc_summary.py <dir> [name-filter]"""
import csv
import glob
import os
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([\w:]+)(<[^(]*>)?', name)
    return (m.group(1).split('::')[-1] + (m.group(2) or '')) if m else name[:60]


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    groups = OrderedDict()
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f, newline='')):
            k = (short(row['Kernel_Name']), row['Grid_Size'], row.get('LDS_Block_Size', ''), row.get('VGPR_Count', ''),
                 row.get('Accum_VGPR_Count', ''))
            if filt and filt not in k[0]:
                continue
            g = groups.setdefault(k, defaultdict(lambda: [0, 0.0]))
            c = g[row['Counter_Name']]
            c[0] += 1
            c[1] += float(row['Counter_Value'])
    for k, g in groups.items():
        n = max(v[0] for v in g.values())
        vals = {c: v[1] / v[0] for c, v in g.items()}
        line = '%-52s grid %-8s lds %-6s vgpr %s+%s n=%d' % (k[0][:52], k[1], k[2], k[3], k[4], n)
        print(line)
        wc = vals.get('SQ_WAVE_CYCLES')
        parts = []
        for c in sorted(vals):
            if wc and c.startswith('SQ_') and c != 'SQ_WAVE_CYCLES' and 'BUSY_CYCLES' not in c:
                parts.append('%s=%.3g (%.0f%%)' % (c.replace('SQ_', ''), vals[c], 100 * vals[c] / wc))
            else:
                parts.append('%s=%.4g' % (c.replace('SQ_', ''), vals[c]))
        print('      ' + '  '.join(parts))


if __name__ == '__main__':
    main()
