"""Summarise rocprofv3 PMC passes (one pass per counter: FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json [step kernel]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes for gfx950: both counters are in
KiB; FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced reads, so it is DOUBLED; WRITE_SIZE is
exact for 16-byte-per-lane stores and float atomics.  traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, averaged
per launch of each kernel family (template arguments stripped).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def family(name):
    name = name.replace('(anonymous namespace)::', '')
    name = name.split('(')[0]
    name = re.sub(r'<.*', '', name)
    name = name.replace('void ', '').strip()
    return name.split('::')[-1]


def read_counter(directory, counter):
    per = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no *counter_collection.csv under %s' % directory)
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                fam = family(row['Kernel_Name'])
                per[fam][0] += 1
                per[fam][1] += float(row['Counter_Value'])
    return per


def steady_per_step(directory, counter, step_kernels, skip=2):
    """Counter total of the STEADY-STATE steps / their number: the dispatches are walked in order, a step ends with one of
    `step_kernels`; the first `skip` steps (plan recording, first-use packs and clears) are left out, and so is everything after the
    last step kernel.  -> (total, steps) or (0.0, 0)"""
    rows = []
    for f in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') == counter:
                    rows.append((int(row['Dispatch_Id']), family(row['Kernel_Name']), float(row['Counter_Value'])))
    rows.sort()
    total, steps, cur, seen = 0.0, 0, 0.0, 0
    for _, fam, v in rows:
        cur += v
        if fam in step_kernels:
            seen += 1
            if seen > skip:
                total += cur
                steps += 1
            cur = 0.0
    return total, steps


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fe, wr = read_counter(fetch_dir, 'FETCH_SIZE'), read_counter(write_dir, 'WRITE_SIZE')
    res = {}
    for fam in sorted(set(fe) | set(wr)):
        nf, f = fe.get(fam, [0, 0.0])
        nw, w = wr.get(fam, [0, 0.0])
        f_launch = f / nf if nf else 0.0
        w_launch = w / nw if nw else 0.0
        res[fam] = {'launches_fetch_pass': nf, 'launches_write_pass': nw,
                    'FETCH_SIZE_KiB_per_launch_raw': round(f_launch, 2),
                    'WRITE_SIZE_KiB_per_launch': round(w_launch, 2),
                    'traffic_bytes_per_launch': round((2.0 * f_launch + w_launch) * 1024.0)}
    # ABI-level groups (bench.py times launches per C-ABI entry point: segnb_conv_fprop = image-tile + general kernel)
    groups = {'conv_fprop': ('conv_fprop_kernel', 'conv_fprop_s1x9_kernel', 'conv_fprop_ws_kernel', 'conv_fprop_rw_kernel', 'conv_fprop_c8_kernel', 'conv_roll_kernel'),
              'conv_wgrad': ('conv_wgrad_kernel', 'conv_wgrad_s1x9_kernel')}
    for gname, fams in groups.items():
        nf = sum(fe.get(f, [0, 0.0])[0] for f in fams)
        nw = sum(wr.get(f, [0, 0.0])[0] for f in fams)
        if nf and nw:
            f_launch = sum(fe.get(f, [0, 0.0])[1] for f in fams) / nf
            w_launch = sum(wr.get(f, [0, 0.0])[1] for f in fams) / nw
            res[gname] = {'launches_fetch_pass': nf, 'launches_write_pass': nw,
                          'FETCH_SIZE_KiB_per_launch_raw': round(f_launch, 2),
                          'WRITE_SIZE_KiB_per_launch': round(w_launch, 2),
                          'traffic_bytes_per_launch': round((2.0 * f_launch + w_launch) * 1024.0)}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    # HBM bytes of ONE training step: everything the profiled command moved, divided by the steps it ran (the optimizer
    # kernel runs exactly once per step; warm-up and the host-enqueue measurement are steps like the timed ones)
    step_kernel = sys.argv[4] if len(sys.argv) > 4 else 'sgd_kernel'
    total_f = sum(v[1] for v in fe.values())
    total_w = sum(v[1] for v in wr.values())
    steps_f, steps_w = fe.get(step_kernel, [0, 0.0])[0], wr.get(step_kernel, [0, 0.0])[0]
    if step_kernel == 'sgd_kernel':
        # (r6) a step whose SGD update is fused with the weight pack ends with sgd_ranges_kernel instead (segnb_sgd_pack_pair_multi +
        # segnb_sgd_ranges): every step runs exactly one of the two
        steps_f += fe.get('sgd_ranges_kernel', [0, 0.0])[0]
        steps_w += wr.get('sgd_ranges_kernel', [0, 0.0])[0]
    per_step = None
    if steps_f and steps_w:
        per_step = round((2.0 * total_f / steps_f + total_w / steps_w) * 1024.0)
    # (r6) the figure bench.py quotes: steady-state steps only -- the first two steps of a run (plan recording, first-use weight packs,
    # buffer clears) are not a training step's traffic, and in a short profiled run they were 1/4 of the steps
    marks = (step_kernel, 'sgd_ranges_kernel') if step_kernel == 'sgd_kernel' else (step_kernel,)
    sf, nsf = steady_per_step(fetch_dir, 'FETCH_SIZE', marks)
    sw, nsw = steady_per_step(write_dir, 'WRITE_SIZE', marks)
    all_steps = per_step
    if nsf and nsw:
        per_step = round((2.0 * sf / nsf + sw / nsw) * 1024.0)
    doc = {'kernel_sources_digest': bench.kernel_sources_digest(),      # bench.py reports `traffic` only for this build
           'bytes_per_step': per_step, 'bytes_per_step_all_steps_of_the_run': all_steps, 'steady_steps': nsf,
           'steps_profiled': steps_f, 'step_kernel': step_kernel,
           'note': 'traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B per launch (gfx950 FETCH_SIZE correction x2); '
                   'separate --pmc passes; average over all launches of the kernel family in the profiled command',
           'kernels': res}
    with open(out, 'w') as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    for fam, v in sorted(res.items(), key=lambda kv: -kv[1]['traffic_bytes_per_launch'] * kv[1]['launches_fetch_pass']):
        print('%-40s n=%5d  %10.1f MB/launch' % (fam, v['launches_fetch_pass'], v['traffic_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
