"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV (dev tool).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --no-graph --steps 6 --no-cpu-baseline
    python tools/trace_analyze.py gpurun_out/trace > gpurun_out/timeline.txt

Takes the last complete step (stem_pack_kernel .. next stem_pack_kernel), prints every kernel with its duration
and the idle gap before it, and totals per kernel family.
"""
import csv, glob, os, sys, collections

root = sys.argv[1]
files = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'stem_pack' in r['Kernel_Name']]
# a step has two stem_pack launches (left, right) back to back: keep the first of each pair
firsts = [i for k, i in enumerate(starts) if k == 0 or i - starts[k - 1] > 3]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2      # steps from the end (the last ones may be the roofline pass)
a, b = firsts[-skip - 1], firsts[-skip]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
wall = int(rows[b]['Start_Timestamp']) - t0
busy = 0
fam = collections.defaultdict(lambda: [0, 0.0])
prev_end = t0
print('step wall %.3f ms, %d kernels' % (wall / 1e6, len(step)))
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name']
    short = name.replace('void ', '').replace('srcnn::', '')
    short = short[:short.index('(')] if '(' in short else short
    short = short[:70]
    d = e - s
    busy += d
    fam[short][0] += 1
    fam[short][1] += d
    print('%9.1f us  +%6.1f gap  %7.1f us  grid=%-8s wg=%-4s %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, d / 1e3,
          r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), short))
    prev_end = max(prev_end, e)
print('\nsum of kernel durations %.3f ms (overlapping streams counted twice), wall %.3f ms' % (busy / 1e6, wall / 1e6))
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print('%8.1f us  %4d x  %s' % (d / 1e3, n, k))
