"""One steady-state training step from a rocprofv3 kernel trace (csv): per queue the busy time, the gaps between
consecutive kernels and the time per kernel family, forward and backward apart (the step boundary = sgd_kernel / sgd_ranges_kernel, the
forward / backward boundary = loss_bwd_kernel).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 8 --warmup 3 ...
    python tools/trace_step.py gpurun_out/trace [step index from the end, default 3] [list | full]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def family(name):
    name = name.replace('(anonymous namespace)::', '')
    name = name.split('(')[0]
    name = re.sub(r'<.*', '', name)
    return name.replace('void ', '').strip().split('::')[-1]


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = []
    with open(f, newline='') as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), family(r['Kernel_Name']), r.get('Queue_Id', '0'),
                         r['Kernel_Name'] + '  grid ' + str(r.get('Grid_Size_X', r.get('Grid_Size', '?')))))
    rows.sort()
    sgd = [i for i, r in enumerate(rows) if r[2] in ('sgd_kernel', 'sgd_ranges_kernel')]      # (the step's last kernel, fused or not)
    lo, hi = sgd[-back - 1], sgd[-back]
    step = rows[lo + 1:hi + 1]
    t0, t1 = rows[lo][1], step[-1][1]
    print('step: %.3f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
    lb = [r for r in step if r[2] == 'loss_bwd_kernel'][0][0]
    print('forward (until loss_bwd starts): %.3f ms; backward + optimizer: %.3f ms' % ((lb - t0) / 1e6, (t1 - lb) / 1e6))
    byq = defaultdict(list)
    for r in step:
        byq[r[3]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _, _, _ in rs)
        gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
        pos = [g for g in gaps if g > 0]
        print('queue %s: %d kernels, busy %.3f ms, span %.3f ms, positive gaps %d totalling %.3f ms (median %.1f us)'
              % (q, len(rs), busy / 1e6, (rs[-1][1] - rs[0][0]) / 1e6, len(pos), sum(pos) / 1e6,
                 sorted(pos)[len(pos) // 2] / 1e3 if pos else 0.0))
    # gaps of the busiest queue by the kernels on either side: a gap that repeats behind ONE kernel family is a packet of the
    # queue that is not a kernel (an event record, a wait) or a host-side step -- how the marker of the weight gradient's
    # fork and UNet16's wait for its side stream (11.15) were found
    main = max(byq.values(), key=len)
    cls = defaultdict(lambda: [0, 0])
    for i in range(len(main) - 1):
        g = main[i + 1][0] - main[i][1]
        if g > 2000:
            cls[(main[i][2], main[i + 1][2])][0] += 1
            cls[(main[i][2], main[i + 1][2])][1] += g
    print('-- gaps > 2 us on the busiest queue, by (kernel before, kernel after)')
    for (a, b), (n, t) in sorted(cls.items(), key=lambda kv: -kv[1][1])[:10]:
        print('   %3d x  %8.1f us   after %-32s before %s' % (n, t / 1e3, a, b))
    for phase, sel in (('forward', lambda r: r[0] < lb), ('backward', lambda r: r[0] >= lb)):
        fam = defaultdict(lambda: [0, 0])
        for r in step:
            if sel(r):
                fam[(r[3], r[2])][0] += 1
                fam[(r[3], r[2])][1] += r[1] - r[0]
        print('--', phase)
        for (q, k), (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            print('   q%-3s %-36s n=%3d %8.1f us' % (q, k, n, t / 1e3))
    if len(sys.argv) > 3:
        full = sys.argv[3] == 'full'        # (full: template arguments and grid size as well)
        for s, e, fam_, q, name in step:
            if full:
                m = re.search(r'<(.*)>', name.replace('(anonymous namespace)::', ''))
                fam_ = '%s  <%s>  %s' % (fam_, (m.group(1) if m else '')[:90], name[name.rindex('  grid '):].strip())
            print('%10.1f %8.1f q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, fam_))


if __name__ == '__main__':
    main()
