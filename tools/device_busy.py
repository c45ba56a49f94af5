"""How busy the device was over a run, from a rocprofv3 --kernel-trace CSV (dev tool): union of all kernel intervals (any queue), idle
gaps by size, kernel time per family, and the same restricted to a window of the run (skip the warm-up).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --config 3 --steps 300 ...
    python tools/device_busy.py gpurun_out/trace [--skip 0.3] [--pairs-per-s V]"""
import argparse
import collections
import csv
import glob
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument('root')
ap.add_argument('--skip', type=float, default=0.35, help='fraction of the run (by time) dropped at the front')
ap.add_argument('--tail', type=float, default=0.05, help='fraction dropped at the end')
ap.add_argument('--marker', default='stem_pack', help='kernel that starts a pair (counts pairs in the window)')
args = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(args.root, '**', '*kernel_trace.csv'), recursive=True):
    rows += list(csv.DictReader(open(f)))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
t_lo, t_hi = iv[0][0], max(e for _, e, _ in iv)
a = t_lo + (t_hi - t_lo) * args.skip
b = t_hi - (t_hi - t_lo) * args.tail
win = [(max(s, a), min(e, b), n) for s, e, n in iv if e > a and s < b]
span = b - a
busy, gaps, cur_s, cur_e = 0, [], None, None
depth_time = collections.Counter()
events = sorted([(s, 1) for s, e, _ in win] + [(e, -1) for s, e, _ in win])
depth, last = 0, a
for t, d in events:
    depth_time[min(depth, 8)] += t - last
    last = t
    depth += d
depth_time[0] += b - last
for s, e, _ in win:
    if cur_e is None:
        cur_s, cur_e = s, e
        if s > a:
            gaps.append(s - a)
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
busy += cur_e - cur_s
pairs = sum(1 for s, e, n in win if args.marker in n) / 2.0
fam = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    short = re.sub(r'\(.*', '', n.replace('void ', '').replace('srcnn::', ''))
    short = re.sub(r'<.*', '', short)
    fam[short][0] += 1
    fam[short][1] += e - s
print('window %.1f ms (%.0f %% .. %.0f %% of the run), %d kernels, %.1f pairs -> %.3f ms per pair' % (span / 1e6, args.skip * 100, (1 - args.tail) * 100, len(win), pairs, span / 1e6 / max(pairs, 1)))
print('device busy (any kernel running) %.1f %% of the window; idle %.3f ms per pair in %d gaps' % (busy / span * 100, (span - busy) / 1e6 / max(pairs, 1), len(gaps)))
for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 5e5), (5e5, 1e12)):
    g = [x for x in gaps if lo <= x < hi]
    print('    gaps %6.0f .. %6.0f us: %5d, %.3f ms per pair' % (lo / 1e3, min(hi, 1e9) / 1e3, len(g), sum(g) / 1e6 / max(pairs, 1)))
print('kernels running at once (share of the window): ' + '  '.join('%d: %.1f %%' % (k, v / span * 100) for k, v in sorted(depth_time.items())))
print('kernel time per pair by family (sum of durations, overlapping kernels both count):')
tot = 0
for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:22]:
    print('    %-44s %7.1f launches  %8.1f us' % (k[:44], c / max(pairs, 1), d / 1e3 / max(pairs, 1)))
for k, (c, d) in fam.items():
    tot += d
print('    %-44s %17s  %8.1f us' % ('all kernels', '', tot / 1e3 / max(pairs, 1)))
