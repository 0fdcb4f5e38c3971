"""Summarise a rocprofv3 --pmc run: per (kernel, grid size) in first-seen order, mean of every counter.
    python tools/pmc_summary.py <dir> [name-filter]

When the run also carries a kernel trace (--kernel-trace beside --pmc, as tools/evidence.sh runs it) and the counter set holds
SQ_VALU_MFMA_BUSY_CYCLES, a column MFMA_UTIL is added per kernel:
    SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz)
= the share of the chip's matrix-pipe cycles the launch kept busy (MI355X_MICROARCH.md: the counter counts cycles, 32 per
v_mfma_f32_32x32x16_bf16; 256 CUs x 4 SIMDs; 2.4 GHz is the engine clock ceiling, so the figure is a lower bound when the chip
runs below it).  Durations under the counter pass are the serialised, profiled ones."""
CLOCK_GHZ = 2.4
SIMDS = 256 * 4
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
    dur = {}                                  # Dispatch_Id -> kernel duration in ns (kernel trace of the same run)
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f, newline='')):
            try:
                dur[row['Dispatch_Id']] = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
            except (KeyError, ValueError):
                pass
    seen = set()
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
            did = row.get('Dispatch_Id')
            if did in dur and (k, did) not in seen:
                seen.add((k, did))
                t = g['_duration_ns']
                t[0] += 1
                t[1] += dur[did]
    for k, g in groups.items():
        n = max(v[0] for v in g.values())
        vals = {c: v[1] / v[0] for c, v in g.items()}
        dns = vals.pop('_duration_ns', None)
        line = '%-52s grid %-8s lds %-6s vgpr %s+%s n=%d' % (k[0][:52], k[1], k[2], k[3], k[4], n)
        if dns:
            line += '  %.1f us' % (dns / 1e3)
            mb = vals.get('SQ_VALU_MFMA_BUSY_CYCLES')
            if mb is not None:
                line += '  MFMA_UTIL=%.1f%%' % (100.0 * mb / (SIMDS * dns * CLOCK_GHZ))
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
