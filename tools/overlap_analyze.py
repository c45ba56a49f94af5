"""Concurrency picture of the multi-stream headline from a rocprofv3 --kernel-trace CSV (dev tool).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr3 -o p -- python bench.py --no-pmc --no-cpu-baseline --no-f32-leg --no-3d-leg --steps 40
    python tools/overlap_analyze.py gpurun_out/tr3 > profiles/overlap_r04.txt

Over the steady-state window of the timed region (the densest 40 % of the trace by kernel count): how many kernels run at a
time, how many workgroup slots they could fill between them (sum over running kernels of min(workgroups, 256 CUs x the
kernel's workgroups per CU by LDS / waves), capped), which kernels run ALONE for how long, and per kernel family the mean
duration against the same family's duration in a one-stream trace if one is given as second argument.
"""
import collections
import csv
import glob
import os
import sys


def load(root):
    rows = []
    for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    out = []
    for r in rows:
        name = r['Kernel_Name'].replace('void ', '').replace('srcnn::', '')
        name = name[:name.index('(')] if '(' in name else name
        gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * max(1, int(r.get('Grid_Size_Y', 1) or 1))
        wx = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1)
        lds = int(r.get('LDS_Block_Size', r.get('Group_Segment_Size', 0)) or 0)
        out.append({'name': name[:60], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp']), 'wgs': max(1, gx // max(wx, 1)),
                    'threads': wx, 'lds': lds, 'queue': r.get('Queue_Id', '?')})
    out.sort(key=lambda r: r['s'])
    return out


def slots(k):
    """workgroups of this kernel the chip can hold at once (LDS and wave limits), and how many it wants"""
    per_cu = 8
    if k['lds'] > 0:
        per_cu = min(per_cu, max(1, (160 * 1024) // k['lds']))
    per_cu = min(per_cu, max(1, 2048 // max(k['threads'], 64)))
    return min(k['wgs'], 256 * per_cu), per_cu


def window(rows, frac=0.4):
    n = len(rows)
    a, b = int(n * (0.5 - frac / 2)), int(n * (0.5 + frac / 2))
    return rows[a:b]


def main():
    rows = window(load(sys.argv[1]))
    t0, t1 = rows[0]['s'], max(r['e'] for r in rows)
    ev = []
    for i, r in enumerate(rows):
        ev.append((r['s'], 1, i))
        ev.append((r['e'], -1, i))
    ev.sort()
    active = set()
    last = t0
    by_level = collections.Counter()
    cu_hist = collections.Counter()
    alone = collections.Counter()
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            by_level[min(len(active), 4)] += dt
            # CUs the running kernels could occupy between them (each kernel: workgroups / its workgroups per CU)
            cus = 0.0
            for j in active:
                want, per_cu = slots(rows[j])
                cus += min(256.0, want / per_cu)
            cu_hist[min(int(min(cus, 256.0) // 32), 8)] += dt
            if len(active) == 1:
                alone[rows[next(iter(active))]['name']] += dt
        last = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    wall = t1 - t0
    busy = sum(r['e'] - r['s'] for r in rows)
    print('window: %d kernels, %.3f ms of wall time, sum of kernel durations %.3f ms -> mean concurrency %.2f' % (len(rows), wall / 1e6, busy / 1e6, busy / wall))
    print('time by number of kernels running: ' + '  '.join('%s%d: %.1f %%' % ('>=' if k == 4 else '', k, 100.0 * v / wall) for k, v in sorted(by_level.items())))
    print('time by CUs the running kernels can occupy between them (workgroups / workgroups-per-CU, summed, capped at 256):')
    for k in sorted(cu_hist):
        print('   %3d-%3d CUs: %5.1f %%' % (k * 32, min(k * 32 + 31, 256), 100.0 * cu_hist[k] / wall))
    print('kernels that run ALONE (nothing else on the chip), share of the wall time:')
    for name, v in alone.most_common(12):
        print('   %5.1f %%  %s' % (100.0 * v / wall, name))
    fam = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        fam[r['name']][0] += 1
        fam[r['name']][1] += r['e'] - r['s']
    ref = {}
    if len(sys.argv) > 2:
        for r in window(load(sys.argv[2])):
            ref.setdefault(r['name'], [0, 0])
            ref[r['name']][0] += 1
            ref[r['name']][1] += r['e'] - r['s']
    print('per kernel family: launches, mean duration here%s' % (' | mean duration in the one-stream trace' if ref else ''))
    for name, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:25]:
        extra = ''
        if name in ref and ref[name][0]:
            extra = ' | %8.1f us  (x%.2f)' % (ref[name][1] / ref[name][0] / 1e3, (d / n) / (ref[name][1] / ref[name][0]))
        print('   %5d x %8.1f us%s  %s' % (n, d / n / 1e3, extra, name))


if __name__ == '__main__':
    main()
