"""Average duration of the conv-engine launches in a rocprofv3 --stats CSV (all template instantiations pooled), to be read
next to bench.py's roofline.avg_launch_ms.  usage: stats_avg.py <kernel_stats.csv>"""
import csv, sys
n = tot = 0
for row in csv.DictReader(open(sys.argv[1])):
    if 'conv_f16s_kernel' in row['Name'] or 'conv_f16x3_kernel' in row['Name'] or 'conv_mfma_kernel' in row['Name']:
        n += int(row['Calls']); tot += float(row['TotalDurationNs'])
print('conv engine launches in the trace: %d, total %.3f ms, average %.2f us per launch' % (n, tot / 1e6, tot / max(n, 1) / 1e3))
