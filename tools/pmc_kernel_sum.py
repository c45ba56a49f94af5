"""Per-launch averages of every counter for kernels whose name contains a pattern (dev tool).
    python tools/pmc_kernel_sum.py <dir> <pattern>"""
import collections, csv, glob, os, sys
tot, n = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
            n[r['Counter_Name']] += 1
for k in sorted(tot):
    print('%-32s per launch %.6g  (%d launches)' % (k, tot[k] / n[k], n[k]))
