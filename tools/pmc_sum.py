"""Sums rocprofv3 --pmc counters per kernel family over the steady-state bench steps (dev tool).

    python tools/pmc_sum.py <dir with pass*/ or <COUNTER>/ sub-directories>  [skip_first_steps]  [traffic.json]
Counts steps by their leading stem_pack_kernel launch(es); the first `skip` steps (autotuning, warm-up) are dropped.
"""
import collections, csv, glob, os, sys

root = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
res = collections.defaultdict(lambda: collections.defaultdict(float))
steps_seen = {}
for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    step, last_pack = -1, -10
    for i, r in enumerate(rows):
        name = r['Kernel_Name']
        if 'stem_pack' in name:
            # a step starts with its stem_pack launch (one for both eyes; two back to back before srcnn_stem_pack_pair)
            if i - last_pack > 3:
                step += 1
            last_pack = i
        if step < skip:
            continue
        short = name.replace('void ', '').replace('srcnn::', '')
        short = short[:short.index('(')] if '(' in short else short
        fam = 'conv engine' if short.startswith('conv_') else ('splitk_reduce' if 'splitk' in short else 'other')
        res[r['Counter_Name']][fam] += float(r['Counter_Value'])
        res[r['Counter_Name']]['all kernels'] += float(r['Counter_Value'])
    steps_seen[f] = step + 1 - skip
    for c in set(r['Counter_Name'] for r in rows):
        res[c]['_steps'] = max(res[c]['_steps'], step + 1 - skip)
for c in sorted(res):
    n = max(res[c]['_steps'], 1)
    print('%-28s (%d steps): ' % (c, n) + '  '.join('%s=%.6g' % (k, v / n) for k, v in sorted(res[c].items()) if k != '_steps'))

if len(sys.argv) > 3 and 'FETCH_SIZE' in res and 'WRITE_SIZE' in res:
    import json
    nf, nw = max(res['FETCH_SIZE']['_steps'], 1), max(res['WRITE_SIZE']['_steps'], 1)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_hash                        # the stamp bench.py checks before it trusts this file
    json.dump({'csrc_sha256': csrc_hash(),
               'conv_fetch_size_kb_per_step': res['FETCH_SIZE']['conv engine'] / nf,
               'conv_write_size_kb_per_step': res['WRITE_SIZE']['conv engine'] / nw,
               'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, summed over the conv-engine launches of one '
                       'bench step (bench.py --streams 1, plans preloaded); FETCH_SIZE still uncorrected here'},
              open(sys.argv[3], 'w'))
